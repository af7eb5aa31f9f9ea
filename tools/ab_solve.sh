# A/B of the solve kernel's second walk as a 256-thread matrix-vector product + one-batch table fill (round 5) against the
# previous form (variants/libstv_hip_oldsolve.so), alternating, one box; phase stamps of the new form first
set -e
cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants
STV_LIB_PATH=$V/libstv_hip_stamps.so python tools/solve_stamps.py 2>/dev/null
for r in 1 2 3; do
  for S in 512 1024; do
    echo -n "old solve "; STV_LIB_PATH=$V/libstv_hip_oldsolve.so python tools/step_time.py $S 300 2>/dev/null | grep "^size"
    echo -n "new solve "; python tools/step_time.py $S 300 2>/dev/null | grep "^size"
  done
done
