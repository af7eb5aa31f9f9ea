set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
S=${1:-512}
export STV_NEXT_W=0
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nw0_$S -- python3 tools/step_time.py $S 60 > /dev/null 2> $R/gpurun_out/nw0_$S.err
python tools/gap_report.py $R/gpurun_out/nw0_$S > $R/gpurun_out/nw0_timeline_$S.txt
export STV_NEXT_W=1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/nw1_$S -- python3 tools/step_time.py $S 60 > /dev/null 2> $R/gpurun_out/nw1_$S.err
python tools/gap_report.py $R/gpurun_out/nw1_$S > $R/gpurun_out/nw1_timeline_$S.txt
find $R/gpurun_out/nw0_$S $R/gpurun_out/nw1_$S -name "*kernel_trace.csv" -delete || true
tail -1 $R/gpurun_out/nw0_timeline_$S.txt $R/gpurun_out/nw1_timeline_$S.txt
