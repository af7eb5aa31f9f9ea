# A/B of the pooling epilogue's register moves in the weight-stationary forward kernel: DPP with bound_ctrl (no "old" value
# to set up) and the row stores behind the pooling chunks (their lane swaps may then overwrite the packed words) against the
# kernel of the previous commit (variants/libstv_hip_headws.so), alternating, one box.
set -e
cd $GRAFT_REPO_ROOT
L=gpurun_out/ws_epilogue_ab.log
: > $L
V=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants/libstv_hip_headws.so
for r in 1 2 3; do
  for H in 1024 512; do
    echo "== before" >> $L; WS_H=$H STV_LIB_PATH=$V python tools/ws_probe.py 2>/dev/null | grep "relu=True" >> $L
    echo "== moves removed" >> $L; WS_H=$H python tools/ws_probe.py 2>/dev/null | grep "relu=True" >> $L
  done
done
for r in 1 2 3; do
  for S in 1024 512; do
    echo -n "before   " >> $L; STV_LIB_PATH=$V python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
    echo -n "after    " >> $L; python tools/step_time.py $S 300 2>/dev/null | grep "^size" >> $L
  done
done
cat $L
