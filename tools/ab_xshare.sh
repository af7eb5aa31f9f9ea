# A/B of ConvArgs::xshare (XCDs per spatial tile: 1 / 2 / 4), alternating, one box
set -e
cd $GRAFT_REPO_ROOT
for g in 2 4; do
  STV_CONV_XSHARE=$g python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "tile or conv" 2>&1 | tail -1
done
for r in 1 2 3; do
  for g in 1 2 4; do
    for S in 512 1024; do
      echo -n "xshare=$g "; STV_CONV_XSHARE=$g python tools/step_time.py $S 300 2>/dev/null | grep "^size" | cut -c1-60
    done
  done
done
