#!/bin/bash
# Debug build whose vq_screen_pipe_kernel keeps s_memtime stamps of its first 96 iterations (wave 0 of each group) in spare LDS.
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps_pipe; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; rm -rf $W/csrc/_obj; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/vq_screen.hip" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b,cnt=1):
    global s
    assert a in s, a
    s=s.replace(a,b,cnt)
T='__builtin_amdgcn_s_memtime()'
rep('namespace {\n','__device__ unsigned long long* g_pipe = nullptr;\nextern "C" int lbm_debug_pipe(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe), &p, sizeof(p)); }\n\nnamespace {\n')
rep('    for (int i = 0; i < I; ++i) {\n        const int buf = i & 1;\n',
    '    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + kSLds + 512);\n    if (threadIdx.x < 384) stamps[threadIdx.x] = 0;\n    for (int i = 0; i < I; ++i) {\n        const int buf = i & 1;\n        const unsigned long long t0 = %s;\n        __builtin_amdgcn_sched_barrier(0);\n' % T)
rep('        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        __syncthreads();\n        if (++ph == P) {',
    '        __builtin_amdgcn_sched_barrier(0);\n        const unsigned long long t1 = %s;\n        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n        __syncthreads();\n        if (++ph == P) {' % T)
rep('        __syncthreads();\n        if (++ph == P) {',
    '        const unsigned long long t2 = %s;\n        if (lane == 0 && wg == 0 && i < 64) { stamps[(g * 64 + i) * 3] = t0; stamps[(g * 64 + i) * 3 + 1] = t1; stamps[(g * 64 + i) * 3 + 2] = t2; }\n        __syncthreads();\n        if (++ph == P) {' % T)
rep('        tile = ntile;\n        slot = nslot;\n    }\n}\n', '        tile = ntile;\n        slot = nslot;\n    }\n    __syncthreads();\n    if (g_pipe != nullptr && threadIdx.x < 384) g_pipe[(size_t)blockIdx.x * 384 + threadIdx.x] = stamps[threadIdx.x];\n}\n')
rep('hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, kSLds + 256)', 'hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, kSLds + 4096)')
rep('hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(kSThreads), kSLds + 256, st,', 'hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(kSThreads), kSLds + 4096, st,')
import os
v=os.environ.get("VARIANT","")
if "nofold" in v:
    rep('const bool folds = hf == 1 || !first;','const bool folds = false;')
if "nolds" in v:
    rep('if (t + kPF < 32) load_t(t + kPF);','if (t + kPF < 2) load_t(t + kPF);')
if "nocopy" in v:
    rep('if (cp && (t & 1) && t < 16) copy_row(ntile, buf ^ 1, t >> 1);','')
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_pipe${VARIANT:+_$VARIANT}.so csrc/*.o
echo built
