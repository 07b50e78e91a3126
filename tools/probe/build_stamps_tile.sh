#!/bin/bash
# Debug build whose conv_tile_kernel accumulates s_memtime deltas per phase of the chunk loop (wave 0 of each block).
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
W=/tmp/lbm_stamps_tile; rm -rf $W; mkdir -p $W/include; cp -r $ROOT/lbm_amd/csrc $W/; rm -rf $W/csrc/_obj; cp $ROOT/include/lbm_abi.h $W/include/
sed -i 's#../../include/lbm_abi.h#../include/lbm_abi.h#' $W/csrc/common.h
python3 - "$W/csrc/conv.hip" <<'PY'
import sys
p=sys.argv[1]; s=open(p).read()
def rep(a,b):
    global s
    assert a in s, a
    s=s.replace(a,b,1)
T='__builtin_amdgcn_s_memtime()'
rep('namespace {\n\nusing f32x16','__device__ unsigned long long* g_tile = nullptr;\nextern "C" int lbm_debug_tile(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tile), &p, sizeof(p)); }\n\nnamespace {\n\nusing f32x16')
rep('    using Cfg = ConvCfg<KS, STRIDE, BM>;\n    constexpr int CK','    const unsigned long long W0 = %s;\n    unsigned long long dA = 0, dB = 0, dC = 0, dD = 0, dE = 0;\n    using Cfg = ConvCfg<KS, STRIDE, BM>;\n    constexpr int CK' % T)
old='''    auto chunk_step = [&](int ch) {
        const int buf = ch & 1;
        mfma_block(buf, 0, QH);
        store_chunk(buf ^ 1, ch + 1);
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);   // the tail re-reads the last chunk (unused)
        mfma_block(buf, QH, QN);
        __syncthreads();
    };
    if (nchunks > 1) {
        chunk_step(0);
        for (int ch = 1; ch + 1 < nchunks; ++ch) chunk_step(ch);
    }
    mfma_block((nchunks - 1) & 1, 0, QN);
'''
new='''    const unsigned long long W1 = %(T)s;
    auto chunk_step = [&](int ch) {
        const int buf = ch & 1;
        const unsigned long long t0 = %(T)s;
        mfma_block(buf, 0, QH);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = %(T)s;
        store_chunk(buf ^ 1, ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t2 = %(T)s;
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t3 = %(T)s;
        mfma_block(buf, QH, QN);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t4 = %(T)s;
        __syncthreads();
        const unsigned long long t5 = %(T)s;
        dA += t1 - t0; dB += t2 - t1; dC += t3 - t2; dD += t4 - t3; dE += t5 - t4;
    };
    if (nchunks > 1) {
        chunk_step(0);
        for (int ch = 1; ch + 1 < nchunks; ++ch) chunk_step(ch);
    }
    const unsigned long long W2 = %(T)s;
    mfma_block((nchunks - 1) & 1, 0, QN);
    asm volatile("" : "+v"(acc[0][0][0]));
    const unsigned long long W3 = %(T)s;
''' % dict(T=T)
rep(old,new)
idx=s.index('// ------------------------------------------------------------------------------------- Winograd F(2,3) path')
k=s.rfind('}\n',0,idx)
s=s[:k]+'    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    const unsigned long long W4 = %s;\n    if (g_tile != nullptr && threadIdx.x == 0) {\n        unsigned long long* q = g_tile + 10 * (size_t)blockIdx.x;\n        q[0] = W1 - W0; q[1] = dA; q[2] = dB; q[3] = dC; q[4] = dD; q[5] = dE; q[6] = W3 - W2; q[7] = W4 - W3; q[8] = W0; q[9] = W4;\n    }\n' % T + s[k:]
open(p,'w').write(s)
PY
cd $W
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -munsafe-fp-atomics -c $f -o ${f%.hip}.o 2>/dev/null & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/tools/probe/liblbm_stamps_tile.so csrc/*.o
echo built
