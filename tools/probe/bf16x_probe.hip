// Probe: how accurate is an fp32 dot product rebuilt from bf16 pieces on the bf16 MFMA, next to the fp32 MFMA chain?
//   a = a1 + a2 + a3 (each piece the bf16 round-to-nearest of the running residual; residuals are exact in fp32)
//   x3: a1b1 + a1b2 + a2b1            x6: x3 + a1b3 + a2b2 + a3b1            x9: all nine
// C[32x32] = A[32xK] * B[Kx32], K = 256, one wavefront, float64 truth on the host.
// usage: bf16x_probe [relu]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 256;

__device__ inline void split3(const float* v, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h1 = (__bf16)v[e];
        const float r1 = v[e] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        p1[e] = h1; p2[e] = h2; p3[e] = (__bf16)r2;
    }
}

// out: [5][32][32]: 0 = fp32 chain, 1 = x3, 2 = x6 (one accumulator, small terms first), 3 = x6 (three accumulators), 4 = x9
__global__ __launch_bounds__(64) void probe(const float* A, const float* B, float* out) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    f32x16 c32, c3, c6, c6h, c6m, c6l, c9;
    for (int r = 0; r < 16; ++r) c32[r] = c3[r] = c6[r] = c6h[r] = c6m[r] = c6l[r] = c9[r] = 0.f;
    for (int t = 0; t < K / 2; ++t)
        c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[j * K + 2 * t + h], B[(2 * t + h) * 32 + j], c32, 0, 0, 0);
    for (int t = 0; t < K / 16; ++t) {
        float av[8], bv[8];
        for (int e = 0; e < 8; ++e) {
            av[e] = A[j * K + 16 * t + 8 * h + e];
            bv[e] = B[(16 * t + 8 * h + e) * 32 + j];
        }
        bf16x8 a1, a2, a3, b1, b2, b3;
        split3(av, a1, a2, a3);
        split3(bv, b1, b2, b3);
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
        MF(a2, b1, c3); MF(a1, b2, c3); MF(a1, b1, c3);
        MF(a3, b1, c6); MF(a2, b2, c6); MF(a1, b3, c6); MF(a2, b1, c6); MF(a1, b2, c6); MF(a1, b1, c6);
        MF(a3, b1, c6l); MF(a2, b2, c6l); MF(a1, b3, c6l); MF(a2, b1, c6m); MF(a1, b2, c6m); MF(a1, b1, c6h);
        MF(a3, b3, c9); MF(a3, b2, c9); MF(a2, b3, c9);
        MF(a3, b1, c9); MF(a2, b2, c9); MF(a1, b3, c9); MF(a2, b1, c9); MF(a1, b2, c9); MF(a1, b1, c9);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[0 * 1024 + row * 32 + j] = c32[r];
        out[1 * 1024 + row * 32 + j] = c3[r];
        out[2 * 1024 + row * 32 + j] = c6[r];
        out[3 * 1024 + row * 32 + j] = (c6l[r] + c6m[r]) + c6h[r];
        out[4 * 1024 + row * 32 + j] = c9[r];
    }
}

// speed: per wave a 64x64 output tile (2x2 fragments), operands from registers, NP products per k-block
template <int NP>
__global__ __launch_bounds__(256) void speed(float* out, int iters) {
    f32x16 acc[2][2];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a >> 1][a & 1][r] = 0.f;
    bf16x8 a[2][3], b[2][3];
    for (int i = 0; i < 2; ++i)
        for (int p = 0; p < 3; ++p)
            for (int e = 0; e < 8; ++e) {
                a[i][p][e] = (__bf16)(0.01f * (threadIdx.x + e + p + i));
                b[i][p][e] = (__bf16)(0.02f * (threadIdx.x - e + p - i));
            }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jn = 0; jn < 2; ++jn) {
                constexpr int PA[9] = {2, 1, 0, 1, 0, 0, 2, 2, 1}, PB[9] = {0, 1, 2, 0, 1, 0, 2, 1, 2};
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA[p]], b[jn][PB[p]], acc[i][jn], 0, 0, 0);
            }
        asm volatile("" : "+v"(a[0][0]), "+v"(b[0][0]));
    }
    float s = 0.f;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q >> 1][q & 1][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const bool relu = argc > 1 && !strcmp(argv[1], "relu");
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    auto nrm = [] {
        double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
        return sqrt(-2 * log(u)) * cos(6.283185307179586 * v);
    };
    for (auto& v : A) { double t = nrm(); v = (float)(relu ? fmax(t, 0.0) : t); }
    for (auto& v : B) v = (float)(nrm() * (relu ? 0.05 : 1.0));
    float *dA, *dB, *dO;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dO, 5 * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dO);
    std::vector<float> O(5 * 1024);
    if (hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    const char* names[5] = {"fp32 mfma chain", "bf16 x3", "bf16 x6 (1 acc)", "bf16 x6 (3 acc)", "bf16 x9"};
    const double eps = ldexp(1.0, -24);
    for (int v = 0; v < 5; ++v) {
        double mx = 0, ss = 0, mxr = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double t = 0, m = 0;
                for (int k = 0; k < K; ++k) { t += (double)A[i * K + k] * B[k * 32 + j]; m += fabs((double)A[i * K + k] * B[k * 32 + j]); }
                const double e = fabs(O[v * 1024 + i * 32 + j] - t);
                mx = fmax(mx, e / m); ss += (e / m) * (e / m);
                mxr = fmax(mxr, e / fmax(fabs(t), 1e-30));
            }
        printf("%-18s err/sum|ab|: max %.3f eps  rms %.3f eps   max err/|c| %.3g\n", names[v], mx / eps, sqrt(ss / 1024) / eps, mxr);
    }
    // speed
    float* dS; hipMalloc(&dS, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000, blocks = 256 * 2;
    for (int np : {3, 6, 9}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (np == 3) hipLaunchKernelGGL(speed<3>, dim3(blocks), dim3(256), 0, 0, dS, iters);
            if (np == 6) hipLaunchKernelGGL(speed<6>, dim3(blocks), dim3(256), 0, 0, dS, iters);
            if (np == 9) hipLaunchKernelGGL(speed<9>, dim3(blocks), dim3(256), 0, 0, dS, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fp32_equiv = 2.0 * 64 * 64 * 16 * (double)iters * 4 * blocks;   // useful (fp32-equivalent) flops
        printf("x%d: %.3f ms, %.1f TFLOP/s fp32-equivalent (%.0f bf16 TFLOP/s)\n", np, ms, fp32_equiv / ms / 1e9, fp32_equiv * np / ms / 1e9);
    }
    return 0;
}
