#!/bin/bash
# Debug build whose welch_fold_kernel records per-group phase stamps (wave 0 of each block, first group).
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
rep('        // ---- stage the group\'s rows (contiguous in x) into this wave\'s LDS\n','        const unsigned long long S0 = __builtin_amdgcn_s_memtime();\n        // ---- stage the group\'s rows (contiguous in x) into this wave\'s LDS\n')
rep('        // ---- per-segment means:','        __builtin_amdgcn_s_waitcnt(0); const unsigned long long S1 = __builtin_amdgcn_s_memtime();\n        // ---- per-segment means:')
rep('        // ---- folded pruned DFT, 16 segments per MFMA tile','        __builtin_amdgcn_s_waitcnt(0); const unsigned long long S2 = __builtin_amdgcn_s_memtime();\n        // ---- folded pruned DFT, 16 segments per MFMA tile')
rep('        // ---- segment average (ascending segment order)','        __builtin_amdgcn_s_waitcnt(0); const unsigned long long S3 = __builtin_amdgcn_s_memtime();\n        // ---- segment average (ascending segment order)')
i=s.index('        // ---- segment average (ascending segment order)')
j=s.index('        __builtin_amdgcn_wave_barrier();\n    }\n}', i)
s=s[:j]+'        __builtin_amdgcn_s_waitcnt(0); const unsigned long long S4 = __builtin_amdgcn_s_memtime();\n        if (g_wstamps != nullptr && lane == 0 && wave == 0 && grp == grp_first + gstride) { unsigned long long* q = g_wstamps + 5 * (size_t)blockIdx.x; q[0]=S0;q[1]=S1;q[2]=S2;q[3]=S3;q[4]=S4; }\n'+s[j:]
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_w.so csrc/*.o
echo built
