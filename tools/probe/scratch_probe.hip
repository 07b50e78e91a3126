// Probe: do a wave's scratch (private segment) slots keep their contents through a loop that also moves tiles by LDS-DMA?
// Background ("spilling LDS-DMA kernel"): a build of the first screening kernel with 13 spilled VGPRs
// (56 bytes of scratch per lane) returned wrong tokens for the upper half-wave (lanes 32-63) of a few waves per launch, run-to-run
// different, only on grids of more than one workgroup per CU; the listing shows every reload correctly waited.  This probe keeps
// per-lane tags in a private array that is forced to scratch (runtime index), re-reads them every iteration and logs every mismatch.
//   hipcc -O3 --offload-arch=gfx950 [-DDMA=0|1] [-DNSLOT=14] [-DBUILTIN=1] -o scratch_probe tools/probe/scratch_probe.hip && ./scratch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef DMA
#define DMA 1
#endif
#ifndef NSLOT
#define NSLOT 14
#endif
#ifndef BUILTIN
#define BUILTIN 1
#endif
#ifndef CONSTIDX
#define CONSTIDX 0
#endif
#ifndef LDSBYTES
#define LDSBYTES 57344
#endif
extern __shared__ __align__(16) unsigned char smem[];

struct Bad { unsigned block, tid, slot, iter, got, want, hwid, xcc; };

__device__ __forceinline__ unsigned tag(unsigned block, unsigned tid, unsigned slot) { return (block * 2654435761u) ^ (tid << 8) ^ (slot * 0x9e3779b9u) ^ 0x5a5a0000u; }

__global__ __launch_bounds__(256, 2) void probe(const unsigned char* __restrict__ tab, size_t tab_bytes, int iters, const int* __restrict__ perm,
                                                Bad* __restrict__ bad, int* __restrict__ nbad, int max_bad, float* __restrict__ sink) {
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if CONSTIDX
    // the instruction form of a register spill (scratch_store_dword off, v, off offset:N), issued by hand inside the frame of a
    // private array that exists only to make the kernel own 4 * NSLOT (rounded) bytes of scratch per lane
    volatile unsigned priv[NSLOT];
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) priv[s] = 0;                                          // every element touched: the frame keeps its size
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#define SSTORE(OFF, V) asm volatile("scratch_store_dword off, %0, off offset:" #OFF :: "v"(V) : "memory")
#define SLOAD(OFF, V) asm volatile("scratch_load_dword %0, off, off offset:" #OFF "\n\ts_waitcnt vmcnt(0)" : "=v"(V) :: "memory")
    SSTORE(0, tag(blockIdx.x, tid, 0)); SSTORE(4, tag(blockIdx.x, tid, 1)); SSTORE(8, tag(blockIdx.x, tid, 2)); SSTORE(12, tag(blockIdx.x, tid, 3));
    SSTORE(16, tag(blockIdx.x, tid, 4)); SSTORE(20, tag(blockIdx.x, tid, 5)); SSTORE(24, tag(blockIdx.x, tid, 6)); SSTORE(28, tag(blockIdx.x, tid, 7));
    SSTORE(32, tag(blockIdx.x, tid, 8)); SSTORE(36, tag(blockIdx.x, tid, 9)); SSTORE(40, tag(blockIdx.x, tid, 10)); SSTORE(44, tag(blockIdx.x, tid, 11));
    SSTORE(48, tag(blockIdx.x, tid, 12)); SSTORE(52, tag(blockIdx.x, tid, 13));
#else
    unsigned priv[NSLOT];
    for (int s = 0; s < NSLOT; ++s) priv[perm[s]] = tag(blockIdx.x, tid, perm[s]);       // runtime index: the array lives in scratch
#endif
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#if DMA
        for (int r = 0; r < 4; ++r) {
            const size_t off = (((size_t)blockIdx.x * iters + it) * 16 + wave * 4 + r) * 1024 % (tab_bytes - 1024);
            const unsigned char* src = tab + (off & ~(size_t)15) + lane * 16;
            unsigned char* dst = smem + ((it & 1) * 16 + wave * 4 + r) * 1024;
#if BUILTIN
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
#else
            const unsigned d32 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)dst);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(d32) : "memory");
#endif
        }
#endif
        // some LDS reads + arithmetic (stand-in for the MFMA stage)
        const float* f = reinterpret_cast<const float*>(smem + ((it & 1) ^ 1) * 16384);
        for (int q = 0; q < 8; ++q) acc = fmaf(f[(tid * 4 + q * 1031) & 4095], 1.0001f, acc);
#if CONSTIDX
        int s = 0;
        unsigned got = 0, want = 0;
        unsigned g_[14];
        SLOAD(0, g_[0]); SLOAD(4, g_[1]); SLOAD(8, g_[2]); SLOAD(12, g_[3]); SLOAD(16, g_[4]); SLOAD(20, g_[5]); SLOAD(24, g_[6]);
        SLOAD(28, g_[7]); SLOAD(32, g_[8]); SLOAD(36, g_[9]); SLOAD(40, g_[10]); SLOAD(44, g_[11]); SLOAD(48, g_[12]); SLOAD(52, g_[13]);
#pragma unroll
        for (int q = 0; q < 14; ++q) {                                                    // every slot, every iteration
            const unsigned w_ = tag(blockIdx.x, tid, q);
            if (g_[q] != w_) { s = q; got = g_[q]; want = w_; }
        }
#else
        const int s = perm[it % NSLOT];                                                   // re-read one slot per iteration (runtime index)
        const unsigned got = priv[s], want = tag(blockIdx.x, tid, s);
#endif
        if (got != want) {
            const int p = atomicAdd(nbad, 1);
            if (p < max_bad) {
                unsigned hw, xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                bad[p] = Bad{blockIdx.x, tid, (unsigned)s, (unsigned)it, got, want, hw, xcc};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (acc == 12345.678f) sink[tid] = acc;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 24000, iters = argc > 2 ? atoi(argv[2]) : 32, reps = argc > 3 ? atoi(argv[3]) : 4;
    const size_t tab_bytes = 64u << 20;
    unsigned char* tab; Bad* bad; int *nbad, *perm; float* sink;
    hipMalloc(&tab, tab_bytes); hipMemset(tab, 1, tab_bytes);
    hipMalloc(&bad, 4096 * sizeof(Bad)); hipMalloc(&nbad, 4); hipMalloc(&perm, NSLOT * 4); hipMalloc(&sink, 1024);
    std::vector<int> hp(NSLOT);
    for (int i = 0; i < NSLOT; ++i) hp[i] = (i * 5 + 3) % NSLOT;                        // a permutation when gcd(5, NSLOT) = 1
    hipMemcpy(perm, hp.data(), NSLOT * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, LDSBYTES);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(probe));
    printf("DMA=%d BUILTIN=%d NSLOT=%d LDS=%d regs=%d scratch(localSizeBytes)=%zu blocks=%d iters=%d\n", DMA, BUILTIN, NSLOT, LDSBYTES, fa.numRegs,
           (size_t)fa.localSizeBytes, blocks, iters);
    for (int rep = 0; rep < reps; ++rep) {
        hipMemset(nbad, 0, 4);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(256), LDSBYTES, 0, tab, tab_bytes, iters, perm, bad, nbad, 4096, sink);
        hipError_t e = hipDeviceSynchronize();
        int n = 0; hipMemcpy(&n, nbad, 4, hipMemcpyDeviceToHost);
        printf("rep %d: %s, %d mismatching (lane, iteration) reads\n", rep, hipGetErrorString(e), n);
        std::vector<Bad> hb(n < 4096 ? n : 4096);
        if (!hb.empty()) hipMemcpy(hb.data(), bad, hb.size() * sizeof(Bad), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < hb.size() && i < 24; ++i)
            printf("   block %u tid %u (wave %u lane %u) slot %u iter %u got %08x want %08x  hw_id %08x (simd %u cu %u sh %u se %u) xcc %u\n", hb[i].block, hb[i].tid,
                   hb[i].tid >> 6, hb[i].tid & 63, hb[i].slot, hb[i].iter, hb[i].got, hb[i].want, hb[i].hwid, (hb[i].hwid >> 4) & 3, (hb[i].hwid >> 8) & 15,
                   (hb[i].hwid >> 12) & 1, (hb[i].hwid >> 13) & 7, hb[i].xcc & 15);
    }
    return 0;
}
