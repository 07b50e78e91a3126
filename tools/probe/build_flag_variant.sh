#!/bin/bash
# Experiment build: tools/probe/liblbm_<NAME>.so = the in-tree library with csrc/<FILE> recompiled with the extra flags in $EXTRA.
# Usage: EXTRA="-DLBM_WS_OUT_HOOK=5" build_flag_variant.sh conv_wino_ws.hip ws_out5 ; run with LBM_AMD_LIB=tools/probe/liblbm_ws_out5.so
set -e
FILE=$1; NAME=$2
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_fv_$NAME; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
cd $W
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics ${EXTRA} -c csrc/$FILE -o csrc/_obj/${FILE%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_$NAME.so csrc/_obj/*.o
echo built $NAME
