#!/bin/bash
# (STAMPS= in the environment builds without the stamps: plain timing of an ablation.)
# Debug builds of the library with conv_wino3_ws_kernel instrumented: -DLBM_WS_STAMPS (s_memtime deltas per phase of its tile walk) and,
# with an argument N, -DLBM_WS_ABLATE=N (1: no output stores, 2: no window copies after the first epoch, 3: both) -> liblbm_stamps_ws[N].so
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps_ws$1; rm -rf $W; mkdir -p $W
EXTRA=""; [ -n "$1" ] && EXTRA="-DLBM_WS_ABLATE=$1"
for f in $ROOT/lbm_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ $b = conv_wino_ws ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics ${STAMPS--DLBM_WS_STAMPS} $EXTRA -c $f -o $W/$b.o 2>/dev/null
  else cp $ROOT/lbm_amd/csrc/_obj/$b.o $W/$b.o; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_ws$1.so $W/*.o
echo built
