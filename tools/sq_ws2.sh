# SQ counters of the forward weight-stationary kernel, one wave per SIMD (conv_ws.hip) against two (conv_ws2.hip).
# usage (GPU box): bash tools/sq_ws2.sh && python tools/summarize_sq.py r05ws2 gpurun_out/sqw2_*
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
export STV_CONV_WS=2
STV_CONV_WS2=0 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sqw2_one_wave -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sqw2_one_wave.err
echo "one wave done"
STV_CONV_WS2=2 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sqw2_two_waves -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sqw2_two_waves.err
echo "two waves done"
C2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
STV_CONV_WS2=0 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $R/gpurun_out/sqw2b_one_wave -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sqw2b_one_wave.err
STV_CONV_WS2=2 rocprofv3 --pmc $C2 --kernel-trace --output-format csv -d $R/gpurun_out/sqw2b_two_waves -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sqw2b_two_waves.err
echo "inst mix done"
find $R/gpurun_out/sqw2* -name "*kernel_trace.csv" -delete || true
