#!/bin/bash
# Experiment build: tools/probe/liblbm_kw_<NAME>.so = the in-tree library with conv_k4s2_wino_ws.hip compiled with -DLBM_KW_ABLATE=<N>
# (1 no stores, 2 no window copies after the first epoch, 4 one operand read per phase, 8 no exchange, 16 stores to consecutive addresses)
# and the extra flags in $EXTRA (e.g. -DLBM_KW_OUT_HOOK=5).  Usage: build_kw_abl.sh NAME N.  Only the launch time is read.
set -e
NAME=$1; N=${2:-0}
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_kw_$NAME; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
cd $W
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -DLBM_KW_ABLATE=$N ${EXTRA} -c csrc/conv_k4s2_wino_ws.hip -o csrc/_obj/conv_k4s2_wino_ws.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_kw_$NAME.so csrc/_obj/*.o
echo built $NAME
