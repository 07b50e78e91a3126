// Microbenchmark: scalar v_fma_f32 vs packed v_pk_fma_f32 throughput on gfx950 (per-CU issue rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f2 = __attribute__((ext_vector_type(2))) float;

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float s) {
    f2 a[8], b = {s, s * 0.5f}, c = {0.25f, 0.125f};
    for (int i = 0; i < 8; ++i) a[i] = (f2){(float)threadIdx.x + i, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {          // 16 scalar FMAs
                a[i].x = __builtin_fmaf(a[i].x, b.x, c.x);
                a[i].y = __builtin_fmaf(a[i].y, b.y, c.y);
            } else {                  // 8 packed FMAs
                a[i] = __builtin_elementwise_fma(a[i], b, c);
            }
        }
    }
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) sum += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 4, iters = 20000;
    const int blocks = 256 * wgs;
    float* out;
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
            else hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 16 * iters * 256.0 * blocks;
            if (rep) printf("mode %d (%s) wgs/cu %d: %.3f ms, %.1f TFLOP/s\n", mode, mode ? "packed" : "scalar", wgs, ms, fl / (ms * 1e-3) / 1e12);
        }
    return 0;
}
