# A/B of the ReLU sweep in the weight-stationary forward kernel (conv_ws.hip, STV_WS_SWEEP): shipped library against
# variants/libstv_hip_nosweep.so (tools/build_variant.sh nosweep "-DSTV_WS_SWEEP=0" conv_ws.hip), alternating, one box
set -e
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants/libstv_hip_nosweep.so
for r in 1 2; do
  echo "== isolated kernel, sweep"; python tools/ws_probe.py 2>/dev/null | grep "relu=True"
  echo "== isolated kernel, per-fragment"; STV_LIB_PATH=$V python tools/ws_probe.py 2>/dev/null | grep "relu=True"
done
for r in 1 2 3; do
  for S in 1024 512; do
    echo -n "sweep      "; python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    echo -n "perfragment "; STV_LIB_PATH=$V python tools/step_time.py $S 300 2>/dev/null | grep "^size"
  done
done
