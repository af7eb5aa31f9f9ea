# A/B of the optimizer update appended to the closure's command buffer (one hipGraph per step, STV_FUSE_STEP=1) against
# four eager launches behind the graph (STV_FUSE_STEP=0), alternating, one box.
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/fuse_step_ab.log
: > $L
timeout -k 10 600 python -m pytest tests/test_gpu_surface.py tests/test_gpu_model.py -x -q -p no:cacheprovider >> $L 2>&1
for r in 1 2 3; do
  for S in 512 1024 256 64; do
    echo -n "eager update " >> $L; STV_FUSE_STEP=0 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "one graph    " >> $L; STV_FUSE_STEP=1 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
  done
done
tail -n 40 $L
