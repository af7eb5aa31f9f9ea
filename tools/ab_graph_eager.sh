# the replayed hipGraph per step against eager launches of the same schedule (STV_HIP_GRAPH=0), alternating, one box
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/graph_eager_ab.log
: > $L
for r in 1 2 3; do
  for S in 512 1024 256; do
    echo -n "graph  " >> $L; STV_HIP_GRAPH=1 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "eager  " >> $L; STV_HIP_GRAPH=0 python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
  done
done
cat $L
