# A/B of the ReLU sweep in the weight-stationary forward kernel (conv_ws.hip, STV_WS_SWEEP): the shipped library (0: ReLU
# per A fragment) against variants/libstv_hip_sweep.so (1: once per staged tile in LDS) and libstv_hip_sweepcounted.so
# (2: the sweep with the counted DMA wait - a timing diagnostic, not a correct kernel) - tools/build_variant.sh NAME
# "-DSTV_WS_SWEEP=n" conv_ws.hip; alternating, one box
set -e
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants
for r in 1 2 3; do
  echo -n "per-fragment   "; python tools/ws_probe.py 2>/dev/null | grep "relu=True"
  echo -n "sweep          "; STV_LIB_PATH=$V/libstv_hip_sweep.so python tools/ws_probe.py 2>/dev/null | grep "relu=True"
  echo -n "sweep, counted "; STV_LIB_PATH=$V/libstv_hip_sweepcounted.so python tools/ws_probe.py 2>/dev/null | grep "relu=True"
done
if [ "$1" = "steps" ]; then
for r in 1 2 3; do
  for S in 1024 512; do
    echo -n "per-fragment "; python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    echo -n "sweep        "; STV_LIB_PATH=$V/libstv_hip_sweep.so python tools/step_time.py $S 300 2>/dev/null | grep "^size"
  done
done
fi
