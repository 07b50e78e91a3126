#!/bin/bash
# Debug build whose fir_fft_kernel records phase stamps (wave 0 of each workgroup, its second item).
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps_f; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; rm -rf $W/csrc/_obj; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/fir_fft.hip" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b):
    global s
    assert s.count(a)==1, (s.count(a), a)
    s=s.replace(a,b,1)
T='__builtin_amdgcn_s_memtime()'
rep('namespace {\n\nstruct FftPlan','__device__ unsigned long long* g_fstamps = nullptr;\nextern "C" int lbm_debug_fstamps(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fstamps), &p, sizeof(p)); }\n\nnamespace {\n\nstruct FftPlan')
def stamp(n): return '        __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long S%d = %s;\n' % (n, T)   # lgkmcnt(0) only
rep('        // ---- real samples -> LDS (sample m sits', '        const unsigned long long S0 = %s;\n        // ---- real samples -> LDS (sample m sits' % T)
rep('        float re[16], im[16];\n#pragma unroll\n        for (int n1 = 0; n1 < 16; ++n1) {\n            const int p = t + Q * n1;', stamp(1) + '        float re[16], im[16];\n#pragma unroll\n        for (int n1 = 0; n1 < 16; ++n1) {\n            const int p = t + Q * n1;')
rep('        // ---- forward stage 1\n', stamp(2) + '        // ---- forward stage 1\n')
rep('        __syncthreads();\n        // ---- forward stage 2\n', stamp(3) + '        __syncthreads();\n' + stamp(4) + '        // ---- forward stage 2\n')
rep('        __syncthreads();   // s_x is reused with the stage-3 layout\n', '        __syncthreads();   // s_x is reused with the stage-3 layout\n' + stamp(5))
rep('        __syncthreads();\n        // ---- forward stage 3, product', stamp(6) + '        __syncthreads();\n' + stamp(7) + '        // ---- forward stage 3, product')
rep('        __syncthreads();\n        // ---- inverse stage 2\n', stamp(8) + '        __syncthreads();\n' + stamp(9) + '        // ---- inverse stage 2\n')
rep('            im[k] = v.y;\n        }\n        __syncthreads();\n#pragma unroll\n        for (int k = 1; k < 16; ++k) {\n            const float2 w = s_t2[k * R3 + n3];\n            const float a = re[k], b = im[k];\n            re[k] = a * w.x + b * w.y;', '            im[k] = v.y;\n        }\n        __syncthreads();\n' + stamp(10) + '#pragma unroll\n        for (int k = 1; k < 16; ++k) {\n            const float2 w = s_t2[k * R3 + n3];\n            const float a = re[k], b = im[k];\n            re[k] = a * w.x + b * w.y;')
rep('        __syncthreads();\n        // ---- inverse stage 1 (reads', stamp(11) + '        __syncthreads();\n' + stamp(12) + '        // ---- inverse stage 1 (reads')
rep('        float* yr = y + row * (long long)T;\n', stamp(13) + '        float* yr = y + row * (long long)T;\n')
rep('                if (b0 + o < T) yr[b0 + o] = im[n1];\n            }\n        }\n', '                if (b0 + o < T) yr[b0 + o] = im[n1];\n            }\n        }\n        { const unsigned long long S14 = %s;\n          if (g_fstamps != nullptr && threadIdx.x == 0 && base == (int)(blockIdx.x * NSUB + 2 * gridDim.x * NSUB)) { unsigned long long* q = g_fstamps + 16 * (size_t)blockIdx.x; q[0]=S0;q[1]=S1;q[2]=S2;q[3]=S3;q[4]=S4;q[5]=S5;q[6]=S6;q[7]=S7;q[8]=S8;q[9]=S9;q[10]=S10;q[11]=S11;q[12]=S12;q[13]=S13;q[14]=S14; } }\n' % T)
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_f.so csrc/*.o
echo built
