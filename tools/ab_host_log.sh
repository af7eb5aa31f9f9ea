# A/B of the loss history ring in host memory written by the combine kernel (a logging point waits for that kernel only:
# STV_HOST_LOG=1) against the ring on the device (a logging point copies behind the whole step: STV_HOST_LOG=0), at the
# reference's logging cadence (every 10 steps), alternating, one box.  LOG_EVERY=100000 = no logging point in the window.
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/host_log_ab.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_surface.py tests/test_gpu_model.py -x -q -p no:cacheprovider >> $L 2>&1
for r in 1 2 3; do
  for S in 512 1024 256; do
    echo -n "device ring  " >> $L; STV_HOST_LOG=0 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "host ring    " >> $L; STV_HOST_LOG=1 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "no log point " >> $L; LOG_EVERY=100000 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
  done
done
tail -n 40 $L
