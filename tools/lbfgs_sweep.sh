# sweep of the L-BFGS sweeps' launch geometry (isolated optimizer step, m = 100), one box
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/lbfgs_sweep.log
: > $L
for S in 512 1024; do
  echo "== size $S: default" >> $L
  python tools/lbfgs_bench.py $S 2>/dev/null >> $L
  for T in 1024 2048 4096; do
    for PG in 1 2 3 4; do
      for PI in 0 1; do
        echo -n "A tile=$T pgroups=$PG pipe=$PI  " >> $L
        STV_LBFGS_TILE=$T STV_LBFGS_PGROUPS=$PG STV_LBFGS_PIPE=$PI python tools/lbfgs_bench.py $S 2>/dev/null >> $L
      done
    done
  done
  for TB in 1024 2048 4096; do
    echo -n "B tile=$TB  " >> $L
    STV_LBFGS_TILE_B=$TB python tools/lbfgs_bench.py $S 2>/dev/null >> $L
  done
  echo "== size $S: default again" >> $L
  python tools/lbfgs_bench.py $S 2>/dev/null >> $L
done
cat $L
