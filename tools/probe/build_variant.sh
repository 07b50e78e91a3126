#!/bin/bash
# Experiment build: liblbm_<name>.so = the in-tree library with conv.hip patched by a python snippet on stdin
# (variable `s` holds the source text).  Usage: build_variant.sh NAME < patch.py ; run with LBM_AMD_LIB=tools/probe/liblbm_NAME.so
set -e
NAME=$1
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_var_$NAME; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
FILE=${2:-conv.hip}
python3 -c "
import sys
p='$W/csrc/$FILE'; s=open(p).read()
exec(sys.stdin.read())
open(p,'w').write(s)
"
cd $W
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -c csrc/$FILE -o csrc/_obj/${FILE%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_$NAME.so csrc/_obj/*.o
echo built $ROOT/tools/probe/liblbm_$NAME.so
