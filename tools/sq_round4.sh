# SQ counters (one pass) of the kernels round 4 is about: the dominant tiles of both sizes, the 16x16x32 tiles beside
# their 32x32x16 twins on the shapes where they differed most, and conv2_2's backward on the weight-stationary kernel
# against the general one.   usage (GPU box): bash tools/sq_round4.sh && python tools/summarize_sq.py r04 gpurun_out/sq4_*
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for spec in "256 256 256 256 0" "256 256 256 256 14" "128 128 512 512 1" "128 128 512 512 13" "128 128 128 256 1" "128 128 128 256 13" "128 128 128 256 5" "64 64 512 512 3" "32 32 512 512 11" "32 32 512 512 16"; do
  set -- $spec
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq4_$1_$2_$3_$4_$5 -- python3 tools/conv_one.py $1 $2 $3 $4 $5 20 > /dev/null 2> $R/gpurun_out/sq4_$1_$2_$3_$4_$5.err
  echo "sq $spec done"
done
# conv2_2 (128 -> 128 at 512^2): forward + pool and backward (mask + Gram term), weight-stationary and general kernel in one pass
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq4_ws128_1024 -- python3 tools/ws128_probe.py 1024 > /dev/null 2> $R/gpurun_out/sq4_ws128_1024.err
echo "sq ws128 done"
# conv1_2 (64 -> 64 at 1024^2) on the weight-stationary kernel: tools/ws_probe.py
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq4_ws64_1024 -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sq4_ws64_1024.err
echo "sq ws64 done"
