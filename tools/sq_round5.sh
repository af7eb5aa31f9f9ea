# SQ counters (one pass of eight) of the kernels round 5 was about: the weight-stationary forward kernel with the ReLU per
# A fragment (shipped) and as a sweep over the staged tile (variants/libstv_hip_sweep.so), and the 64^2 x 512 -> 512 layer
# on its shipped tile and with K split across workgroups.
# usage (GPU box): bash tools/sq_round5.sh && python tools/summarize_sq.py r05 gpurun_out/sq5_*
set -e
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd $R
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq5_ws64_perfragment -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sq5_ws64_perfragment.err
echo "sq ws64 per-fragment done"
STV_LIB_PATH=$R/style_transfer_visualizer_amd/variants/libstv_hip_sweep.so rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq5_ws64_sweep -- python3 tools/ws_probe.py > /dev/null 2> $R/gpurun_out/sq5_ws64_sweep.err
echo "sq ws64 sweep done"
rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/sq5_xk_64_64_512_512 -- python3 tools/xk_probe.py 20 > /dev/null 2> $R/gpurun_out/sq5_xk_64_64_512_512.err
echo "sq xk done"
find $R/gpurun_out/sq5_* -name "*kernel_trace.csv" -delete || true
