# A/B of the register-staged halo tile of the weight-stationary forward kernel behind a ReLU (STV_WS_RSTAGE=1) against
# LDS-DMA + per-fragment clamp (variants/libstv_hip_norstage.so: -DSTV_WS_RSTAGE=0), alternating, one box.
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/ws_rstage_ab.log
: > $L
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -x -q -k "conv_ws" -p no:cacheprovider >> $L 2>&1
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants/libstv_hip_norstage.so
for r in 1 2 3; do
  for H in 1024 512; do
    echo "== LDS-DMA + clamp per fragment" >> $L; WS_H=$H STV_LIB_PATH=$V python tools/ws_probe.py 2>/dev/null | grep "relu=True" >> $L
    echo "== register-staged" >> $L; WS_H=$H python tools/ws_probe.py 2>/dev/null | grep "relu=True" >> $L
  done
done
for r in 1 2 3; do
  for S in 1024 512; do
    echo -n "dma      " >> $L; STV_LIB_PATH=$V python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "rstage   " >> $L; python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
  done
done
tail -n 50 $L
