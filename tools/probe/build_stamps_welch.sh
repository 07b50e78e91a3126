#!/bin/bash
# Debug build whose welch_team_kernel records per-group phase stamps (wave 0 of each block, second group).
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps_w; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; rm -rf $W/csrc/_obj; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/welch.hip" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a
    s=s.replace(a,b,1)
rep('namespace {\n','__device__ unsigned long long* g_wstamps = nullptr;\nextern "C" int lbm_debug_wstamps(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), &p, sizeof(p)); }\n\nnamespace {\n')
T='__builtin_amdgcn_s_memtime()'
rep('        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        __syncthreads();   // B:', '        const unsigned long long S0 = %s;\n        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        const unsigned long long S0b = %s;\n        __syncthreads();   // B:' % (T, T))
rep('        // ---- per-segment means (scipy detrend=\'constant\') -- only where the mean can reach', '        const unsigned long long S1 = %s;\n        // ---- per-segment means (scipy detrend=\'constant\') -- only where the mean can reach' % T)
rep('        __syncthreads();   // C:', '        __builtin_amdgcn_s_waitcnt(0); const unsigned long long S2 = %s;\n        __syncthreads();   // C:' % T)
rep('        // ---- partial transforms of the non-owner waves -> LDS.', '        const unsigned long long S3 = %s;\n        // ---- partial transforms of the non-owner waves -> LDS.' % T)
rep('        __syncthreads();   // D\n', '        __syncthreads();   // D\n        const unsigned long long S4 = %s;\n' % T)
rep('        __syncthreads();   // E: P complete\n', '        __syncthreads();   // E: P complete\n        const unsigned long long S5 = %s;\n' % T)
rep('        // the next group writes the partials after its barrier C', '        const unsigned long long S6 = %s;\n        if (g_wstamps != nullptr && lane == 0 && tw == 0 && grp == (int64_t)blockIdx.x + gridDim.x) { unsigned long long* q = g_wstamps + 8 * (size_t)blockIdx.x; q[0]=S0;q[1]=S0b;q[2]=S1;q[3]=S2;q[4]=S3;q[5]=S4;q[6]=S5;q[7]=S6; }\n        // the next group writes the partials after its barrier C' % T)
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_w.so csrc/*.o
echo built
