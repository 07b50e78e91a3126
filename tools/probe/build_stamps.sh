#!/bin/bash
# Debug build of liblbm_hip.so whose conv tile kernel records s_memtime stamps per block (see stamps.py).
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; rm -rf $W/csrc/_obj; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/conv.hip" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a
    s=s.replace(a,b,1)
rep('namespace {\n\nusing f32x16','__device__ unsigned long long* g_stamps = nullptr;\nextern "C" int lbm_debug_stamps(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)); }\n\nnamespace {\n\nusing f32x16')
rep('    extern __shared__ __align__(16) float smem[];\n    float* const s_a0 = smem;','    const unsigned long long T0 = __builtin_amdgcn_s_memtime();\n    const unsigned long long R0 = __builtin_amdgcn_s_memrealtime();\n    extern __shared__ __align__(16) float smem[];\n    float* const s_a0 = smem;')
rep('    store_chunk(0, 0);\n    __syncthreads();\n','    store_chunk(0, 0);\n    __syncthreads();\n    const unsigned long long T1 = __builtin_amdgcn_s_memtime();\n')
rep('    mfma_block((nchunks - 1) & 1, 0, QN);\n','    mfma_block((nchunks - 1) & 1, 0, QN);\n    asm volatile("" : "+v"(acc[0][0][0]));\n    const unsigned long long T2 = __builtin_amdgcn_s_memtime();\n')
idx=s.index('// ---------------------------------------------------------------------------------------------- generic path')
k=s.rfind('}\n',0,idx)
s=s[:k]+'    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    const unsigned long long T3 = __builtin_amdgcn_s_memtime();\n    if (g_stamps != nullptr && threadIdx.x == 0) {\n        unsigned long long* q = g_stamps + 4 * (size_t)blockIdx.x;\n        q[0] = T0; q[1] = T1; q[2] = T2; q[3] = T3;\n        if (blockIdx.x & 1) { q[1] = R0; q[2] = __builtin_amdgcn_s_memrealtime(); }\n    }\n'+s[k:]
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps.so csrc/*.o
echo built $ROOT/tools/probe/liblbm_stamps.so
