# usage (on the GPU box): bash tools/profile_round.sh [round tag, default r04]
set -e
R=$GRAFT_REPO_ROOT
T=${1:-r04}
export TMPDIR=/tmp
cd $R
for S in 512 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}a_s$S -- python3 bench.py --no-cpu-baseline --no-extra --size $S > $R/gpurun_out/${T}a_bench_$S.json 2> $R/gpurun_out/${T}a_s$S.err
  echo "stats $S done"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}a_f$S -- python3 bench.py --no-cpu-baseline --no-extra --size $S --steps 3 --warmup 2 > /dev/null 2> $R/gpurun_out/${T}a_f$S.err
  echo "fetch $S done"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/${T}a_w$S -- python3 bench.py --no-cpu-baseline --no-extra --size $S --steps 3 --warmup 2 > /dev/null 2> $R/gpurun_out/${T}a_w$S.err
  echo "write $S done"
  python tools/gap_report.py $R/gpurun_out/${T}a_s$S > $R/gpurun_out/${T}_step_timeline_$S.txt
done
python tools/summarize_profile.py $T $R/gpurun_out/${T}a
ls -la profiles | tail -8
# keep only the small outputs
find $R/gpurun_out/${T}a_s512 $R/gpurun_out/${T}a_s1024 $R/gpurun_out/${T}a_f512 $R/gpurun_out/${T}a_f1024 $R/gpurun_out/${T}a_w512 $R/gpurun_out/${T}a_w1024 -name "*kernel_trace.csv" -delete || true
find $R/gpurun_out -name "*counter_collection.csv" -size +5M -delete || true
