#!/bin/bash
# A/B of the two-waves-per-SIMD forward kernel (conv_ws2.hip) against conv_ws.hip: kernel times, then whole steps.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/ws2_ab.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -k "ws2" -p no:cacheprovider >> $L 2>&1
for H in 1024 512; do
  for mode in "0 1" "2 1" "2 0" "0 1" "2 1"; do
    set -- $mode
    WS_H=$H STV_CONV_WS=2 STV_CONV_WS2=$1 STV_WS2_SKEW=$2 timeout -k 10 120 python tools/ws_probe.py 2>&1 | grep "fwd" >> $L
  done
done
for S in 512 1024; do
  for m in 0 1 0 1; do
    echo "STV_CONV_WS2=$m size $S" >> $L
    STV_CONV_WS2=$m timeout -k 10 200 python tools/step_time.py $S 300 >> $L 2>&1
  done
done
tail -n 60 $L
