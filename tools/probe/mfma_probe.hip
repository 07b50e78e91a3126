// Microbenchmark: what fp32 MFMA rate do the conv/VQ inner-loop shapes sustain on gfx950?
//   mode 0: registers only, 4 accumulators, back-to-back v_mfma_f32_32x32x2_f32
//   mode 1: conv inner loop: per 4 k-steps 2 ds_read_b128 (A) + per k-step 2 ds_read_b32 (B), 4 MFMAs per k-step
//   mode 2: as 1 with all LDS reads of a 4-k-step group issued one group ahead (software prefetch)
//   mode 3: VQ shape: B in registers, A via ds_read_b128 per 4 MFMAs, ONE accumulator chain
// usage: mfma_probe <mode> <wgs_per_cu> <iters>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int LDA = 36, XWP = 128;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __align__(16) float s_a[128 * LDA];
    __shared__ __align__(16) float s_x[32 * XWP];
    for (int i = threadIdx.x; i < 128 * LDA; i += 256) s_a[i] = (float)((i * 7) % 13) * 0.01f;
    for (int i = threadIdx.x; i < 32 * XWP; i += 256) s_x[i] = (float)((i * 5) % 11) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const float* ap = s_a + (wm * 64 + j) * LDA + h * 16;
    const float* bp = s_x + (h * 16) * XWP + wn * 64 + j;
    if (MODE == 0) {
        float a0 = ap[0], a1 = ap[32 * LDA], b0 = bp[0], b1 = bp[32];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
            }
        }
    } else if (MODE == 1 || MODE == 2) {
        for (int it = 0; it < iters; ++it) {
            const float* app = ap + (it & 1);   // defeat hoisting
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t0 = *reinterpret_cast<const float4*>(app + 4 * q - (it & 1));
                const float4 t1 = *reinterpret_cast<const float4*>(app + 32 * LDA + 4 * q - (it & 1));
                const float av0[4] = {t0.x, t0.y, t0.z, t0.w}, av1[4] = {t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float b0 = bp[(4 * q + u) * XWP + (it & 1)], b1 = bp[(4 * q + u) * XWP + 32 + (it & 1)];
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u], b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[u], b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u], b0, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[u], b1, acc[3], 0, 0, 0);
                }
            }
        }
    } else {
        float bx[64];
        for (int s = 0; s < 64; ++s) bx[s] = bp[(s & 15) * XWP + (s >> 4)];
        for (int it = 0; it < iters; ++it) {
            const float* app = ap + (it & 1) * 32 * LDA;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t0 = *reinterpret_cast<const float4*>(app + 4 * q);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t0.x, bx[4 * q + 0 + 16 * (it & 3)], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t0.y, bx[4 * q + 1 + 16 * (it & 3)], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t0.z, bx[4 * q + 2 + 16 * (it & 3)], acc[0], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(t0.w, bx[4 * q + 3 + 16 * (it & 3)], acc[0], 0, 0, 0);
            }
        }
    }
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) sum += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0, wgs = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 2000;
    const int blocks = 256 * wgs;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        else hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double mf = (mode == 3 ? 16.0 : 64.0) * iters * 4.0 * blocks;   // MFMAs
        if (rep) printf("mode %d wgs/cu %d: %.3f ms, %.1f TFLOP/s\n", mode, wgs, ms, mf * 4096.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
