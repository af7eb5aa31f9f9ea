# A/B of the K split across workgroups (STV_CONV_XK = 0 / 1): probe of the layer shapes, then whole steps, alternating, one box
set -e
cd $GRAFT_REPO_ROOT
python tools/xk_probe.py 2>/dev/null
for r in 1 2 3; do
  for S in 512 1024 256; do
    for v in 0 1; do
      echo -n "xk=$v "; STV_CONV_XK=$v python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    done
  done
done
