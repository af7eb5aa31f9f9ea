set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16_layerwise.py -q -m gpu -x -k 512 2>&1 | tail -3
for r in 1 2 3; do
  for v in 0 1; do
    for S in 512 1024; do
      echo -n "arena=$v "; STV_GRAD_ARENA=$v python tools/step_time.py $S 300
    done
  done
done
