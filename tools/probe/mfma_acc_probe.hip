// How exact is ONE v_mfma_f32_32x32x16_bf16 (D = C + sum of 16 bf16 products per output)?  The screening quantizer's error bound
// (csrc/vq_screen.hip) prices it at 16 sequential round-to-nearest adds, i.e. <= 2^-20 (|C| + sum |a_i b_i|).  This probe measures the
// real error against float64 over random operands with mixed magnitudes and signs (cancellation included).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_acc_probe mfma_acc_probe.hip && ./mfma_acc_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;

// one wave: A[32][16] (row i = lane & 31, k = 8 (lane >> 5) + e), B[16][32] (col j = lane & 31, same k), C/D 32x32
__global__ void probe(const __bf16* A, const __bf16* B, const float* C, float* D, int trials) {
    const int lane = threadIdx.x;
    for (int t = blockIdx.x; t < trials; t += gridDim.x) {
        bf16x8 a, b;
        for (int e = 0; e < 8; ++e) {
            a[e] = A[((size_t)t * 32 + (lane & 31)) * 16 + 8 * (lane >> 5) + e];
            b[e] = B[((size_t)t * 32 + (lane & 31)) * 16 + 8 * (lane >> 5) + e];
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            c[r] = C[((size_t)t * 32 + row) * 32 + (lane & 31)];
        }
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            D[((size_t)t * 32 + row) * 32 + (lane & 31)] = c[r];
        }
    }
}

static float bf16_round(float v) {   // round to nearest even to 8 significant bits
    uint32_t u;
    memcpy(&u, &v, 4);
    u += 0x7fff + ((u >> 16) & 1);
    u &= 0xffff0000u;
    memcpy(&v, &u, 4);
    return v;
}

int main() {
    const int trials = 4096;
    std::vector<float> a(trials * 32 * 16), b(trials * 32 * 16), c(trials * 32 * 32), d(trials * 32 * 32);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int t = 0; t < trials; ++t) {
        const int mode = t % 8;   // 0: same scale, 1: exponents spread over 2^+-6, 2: heavy cancellation against C, 3: tiny C,
                                  // 4-7: adversarial for an adder that aligns to the largest term and drops low bits: one product of
                                  // 1 (or C = 1) and the others just below 2^-23 / 2^-24 / 2^-25, same sign or alternating
        for (int i = 0; i < 32 * 16; ++i) {
            const double ea = mode == 1 ? std::ldexp(1.0, (int)(rnd() * 13) - 6) : 1.0;
            a[t * 512 + i] = bf16_round((float)((rnd() * 2 - 1) * ea));
            b[t * 512 + i] = bf16_round((float)((rnd() * 2 - 1) * (mode == 1 ? std::ldexp(1.0, (int)(rnd() * 13) - 6) : 1.0)));
        }
        for (int i = 0; i < 32 * 32; ++i) c[t * 1024 + i] = (float)((rnd() * 2 - 1) * (mode == 2 ? 64.0 : (mode == 3 ? 1e-6 : 4.0)));
        if (mode >= 4) {
            const int sh = 23 + (t / 8) % 3;                      // small products of 0.992 * 2^-sh
            for (int i = 0; i < 32; ++i)
                for (int k = 0; k < 16; ++k) {
                    const bool big = (mode == 4 || mode == 6) && k == (i % 16);
                    const float sgn = (mode >= 6 && (k & 1)) ? -1.f : 1.f;
                    a[t * 512 + i * 16 + k] = big ? 1.f : sgn * 0.9921875f * std::ldexp(1.f, -12);
                    b[t * 512 + i * 16 + k] = big ? 1.f : std::ldexp(1.f, -(sh - 12));
                }
            for (int i = 0; i < 32 * 32; ++i) c[t * 1024 + i] = (mode == 5 || mode == 7) ? 1.f : 0.f;
        }
    }
    std::vector<__bf16> ah(a.size()), bh(b.size());
    for (size_t i = 0; i < a.size(); ++i) { ah[i] = (__bf16)a[i]; bh[i] = (__bf16)b[i]; }
    __bf16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, ah.size() * 2); hipMalloc(&dB, bh.size() * 2); hipMalloc(&dC, c.size() * 4); hipMalloc(&dD, d.size() * 4);
    hipMemcpy(dA, ah.data(), ah.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, bh.data(), bh.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, c.data(), c.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, trials);
    hipMemcpy(d.data(), dD, d.size() * 4, hipMemcpyDeviceToHost);
    double worst[8] = {0}, worst_res[8] = {0};
    for (int t = 0; t < trials; ++t)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ex = c[t * 1024 + i * 32 + j], mag = std::fabs(ex);
                for (int k = 0; k < 16; ++k) {
                    const double p = (double)a[t * 512 + i * 16 + k] * (double)b[t * 512 + j * 16 + k];
                    ex += p;
                    mag += std::fabs(p);
                }
                const double err = std::fabs((double)d[t * 1024 + i * 32 + j] - ex);
                worst[t % 8] = std::fmax(worst[t % 8], err / (mag * std::ldexp(1.0, -24)));
                worst_res[t % 8] = std::fmax(worst_res[t % 8], err / (std::fmax(std::fabs(ex), 1e-300) * std::ldexp(1.0, -24)));
            }
    const char* names[8] = {"same scale", "exponents spread", "cancellation against C", "tiny C", "1 + 15 tiny (products)", "C = 1 + 16 tiny",
                            "1 + 15 tiny, alternating", "C = 1 + 16 tiny, alternating"};
    for (int m = 0; m < 8; ++m)
        printf("%-24s max error = %.3f x 2^-24 (|C| + sum|a b|)   [= %.2f ulp of the result]   (bound used: 16)\n", names[m], worst[m], worst_res[m]);
    return 0;
}
