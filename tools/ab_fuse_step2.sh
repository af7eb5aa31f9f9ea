# host-side cost of a step (LOG_EVERY=100000: no flush inside the timed window) with / without the update in the closure's graph
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/fuse_step_host.log
: > $L
for r in 1 2; do
  for S in 512 64; do
    for F in 0 1; do
      for LE in 10 100000; do
        echo -n "fuse=$F log_every=$LE " >> $L; LOG_EVERY=$LE STV_FUSE_STEP=$F python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
      done
    done
  done
done
cat $L
