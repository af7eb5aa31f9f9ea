set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in 512 1024; do
  for F in 0 1; do
    STV_FUSE_STEP=$F rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sg_${S}_fuse$F -- python3 tools/step_time.py $S 100 > /dev/null 2>&1
    python tools/step_gaps.py gpurun_out/sg_${S}_fuse$F >> gpurun_out/step_gaps.log
  done
done
find gpurun_out/sg_* -name "*.csv" -delete || true
cat gpurun_out/step_gaps.log
