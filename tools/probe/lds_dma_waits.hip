// Reproducer: with __builtin_amdgcn_global_load_lds in the loop hipcc (ROCm 7.2) emits s_waitcnt lgkmcnt(0) before every k-step's MFMAs;
// without it (or with the copy issued through inline assembly, -DASM) the waits are counted (lgkmcnt(2)) and the one-step prefetch works.
//   hipcc -O3 --offload-arch=gfx950 [-DDMA | -DDMA -DASM] -S --cuda-device-only -o - tools/probe/lds_dma_waits.hip | grep s_waitcnt | sort | uniq -c
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;
extern __shared__ __align__(16) unsigned char smem[];
__global__ __launch_bounds__(512) void k(const float* x, const unsigned char* tab, float* out, int n) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 X[16];
    for (int s = 0; s < 16; ++s) X[s] = *reinterpret_cast<const bf16x8*>(x + (lane * 16 + s) * 4);
    f32x16 cur = {};
    for (int i = 0; i < n; ++i) {
        const int buf = i & 1;
#if defined(DMA) && defined(ASM)
        for (int r = 0; r < 8; ++r) {
            const unsigned char* src = tab + (i * 64 + wave * 8 + r) * 1024 + lane * 16;
            const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(smem + (buf ^ 1) * 66560 + (wave * 8 + r) * 1040);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
#elif defined(DMA)
        for (int r = 0; r < 8; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tab + (i * 64 + wave * 8 + r) * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(smem + (buf ^ 1) * 66560 + (wave * 8 + r) * 1040), 16, 0, 0);
#endif
        const unsigned char* ap = smem + buf * 66560 + lane * 1040;
        bf16x8 af[2][2];
        auto load_s = [&](int s_, int par) {
            af[par][0] = *reinterpret_cast<const bf16x8*>(ap + 32 * s_);
            af[par][1] = *reinterpret_cast<const bf16x8*>(ap + 512 + 32 * s_);
        };
        load_s(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (s_ + 1 < 16) load_s(s_ + 1, (s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8* f = af[s_ & 1];
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], X[s_], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], X[s_], cur, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    for (int r = 0; r < 16; ++r) out[threadIdx.x * 16 + r] = cur[r];
}
