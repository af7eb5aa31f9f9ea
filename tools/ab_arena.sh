# A/B of the gradient slabs (plan.alloc_grads): STV_GRAD_ARENA=0 (one tensor per node) against the default, alternating, one box
set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_bf16_layerwise.py tests/test_gpu_model.py -q -m gpu -x 2>&1 | tail -3
for r in 1 2 3; do
  for v in 0 1; do
    for S in 512 1024; do
      echo -n "arena=$v "; STV_GRAD_ARENA=$v python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    done
  done
done
