# A/B of the weight-stationary kernel's resident weights pinned to AGPRs (STV_WS_W_AGPR=1, shipped) against the register
# allocator's own placement (variants/libstv_hip_wvgpr.so: -DSTV_WS_W_AGPR=0), alternating, one box
set -e
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants/libstv_hip_wvgpr.so
for r in 1 2 3; do
  echo "== allocator's placement"; STV_LIB_PATH=$V python tools/ws_probe.py 2>/dev/null
  echo "== weights in AGPRs"; python tools/ws_probe.py 2>/dev/null
done
for r in 1 2 3; do
  for S in 1024 512; do
    echo -n "allocator "; STV_LIB_PATH=$V python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    echo -n "agpr      "; python tools/step_time.py $S 300 2>/dev/null | grep "^size"
  done
done
