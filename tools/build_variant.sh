#!/bin/bash
# A side build of libstv_hip.so for an A/B arm: the listed sources are recompiled with extra flags, every other object is
# the main build's.  usage (build container): tools/build_variant.sh NAME "-DSTV_WS_SWEEP=0" conv_ws.hip [more.hip ...]
# -> style_transfer_visualizer_amd/variants/libstv_hip_NAME.so (travels with the gpurun snapshot; select it with
#    STV_LIB_PATH=$GRAFT_REPO_ROOT/style_transfer_visualizer_amd/variants/libstv_hip_NAME.so)
set -e
NAME=$1; FLAGS=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/style_transfer_visualizer_amd/csrc
OUT=$ROOT/style_transfer_visualizer_amd/variants
mkdir -p $OUT/obj_$NAME
make -C $CSRC -j8 > /dev/null
OBJS=""
for src in conv_igemm conv_igemm16 conv_ws conv_ws2 conv_first pointwise gram optim lbfgs_compact program; do
  if [[ " $* " == *" $src.hip "* ]]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable $FLAGS -I$CSRC -c $CSRC/$src.hip -o $OUT/obj_$NAME/$src.o &
    OBJS="$OBJS $OUT/obj_$NAME/$src.o"
  else
    OBJS="$OBJS $CSRC/$src.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $OUT/libstv_hip_$NAME.so
echo "built $OUT/libstv_hip_$NAME.so"
