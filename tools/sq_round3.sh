set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for spec in "32 32 512 512 8" "32 32 512 512 11" "64 64 512 256 7" "64 64 512 256 12" "64 64 512 512 4" "128 128 256 256 1"; do
  set -- $spec
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq3_$1_$2_$3_$4_$5 -- python3 tools/conv_one.py $1 $2 $3 $4 $5 20 > /dev/null 2> $R/gpurun_out/sq3_$1_$2_$3_$4_$5.err
  echo "sq $spec done"
done
