// Conv1d (+bias, +ReLU) of the VQ-VAE encoder (a10) as an implicit GEMM on fp32 MFMA.
//
// y[b, co, p] = bias[co] + sum_{ci,kt} w[co, ci, kt] * x[b, ci, p*stride - pad + kt]
// GEMM view: M = Cout, N = B*Lout (all output positions of the batch), K = Cin*ksize;
// A = w viewed as [Cout][Cin*ksize] (torch's weight layout is already that row-major matrix),
// B[kk][n] gathered on the fly from x (im2col is never materialised).
// Workgroup tile 64 (co) x 128 (positions), 4 wavefronts as 2x2, each 32 x 64 = two 32x32 MFMA
// accumulators; K is consumed in chunks of 32 staged in LDS.  v_mfma_f32_32x32x2_f32 = exact fp32
// fmaf chain in ascending (ci, kt) order within each lane half (h = 0: first 16 of the chunk, h = 1: last 16).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BM = 64, BN = 128, KC = 32;
constexpr int LDA = KC + 4;  // padded: conflict-free ds_read_b128 of A fragments
constexpr int LDB = BN;

__global__ __launch_bounds__(kThreads) void conv1d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          int64_t Ntot, int Cin, int Lin, int Cout, int Lout, int ks,
                                                          int stride, int pad, int relu) {
    __shared__ __align__(16) float s_a[BM * LDA];
    __shared__ __align__(16) float s_b[KC * LDB];

    const int Ktot = Cin * ks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM;
    const int64_t n0 = (int64_t)blockIdx.x * BN;

    // B-tile fill assignment: this thread always fills column nb, rows kb, kb+2, ...
    const int nb = threadIdx.x & (BN - 1);
    const int kb = threadIdx.x >> 7;  // 0 or 1
    const int64_t ncol = n0 + nb;
    const bool col_ok = ncol < Ntot;
    const int64_t bcol = col_ok ? ncol / Lout : 0;
    const int pcol = col_ok ? (int)(ncol - bcol * Lout) : 0;
    const float* xcol = x + bcol * (int64_t)Cin * Lin;
    const int pos0 = pcol * stride - pad;

    // A-tile fill assignment: row am + 8*i, column ak
    const int ak = threadIdx.x & 31, am = threadIdx.x >> 5;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }

    // running (ci, kt) of this thread's first B row in the current chunk
    int ci_run = kb / ks, kt_run = kb - ci_run * ks;

    for (int kk0 = 0; kk0 < Ktot; kk0 += KC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BM / 8; ++i) {
            const int m = am + 8 * i;
            const int co = m0 + m, kk = kk0 + ak;
            s_a[m * LDA + ak] = (co < Cout && kk < Ktot) ? w[(int64_t)co * Ktot + kk] : 0.f;
        }
        {
            int ci = ci_run, kt = kt_run;
#pragma unroll
            for (int i = 0; i < KC / 2; ++i) {
                const int kl = kb + 2 * i;
                float v = 0.f;
                const int pos = pos0 + kt;
                if (col_ok && ci < Cin && pos >= 0 && pos < Lin) v = xcol[(int64_t)ci * Lin + pos];
                s_b[kl * LDB + nb] = v;
                kt += 2;
                while (kt >= ks) {
                    kt -= ks;
                    ++ci;
                }
            }
            ci_run = ci;
            kt_run = kt;
        }
        __syncthreads();

        const float* ap = s_a + (wm * 32 + j) * LDA + h * (KC / 2);
        const float* bp = s_b + (h * (KC / 2)) * LDB + wn * 64 + j;
#pragma unroll
        for (int q = 0; q < KC / 8; ++q) {
            const float4 a = reinterpret_cast<const float4*>(ap)[q];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bp[s * LDB], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bp[s * LDB + 32], acc1, 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int64_t n = n0 + wn * 64 + sub * 32 + j;
        if (n >= Ntot) continue;
        const int64_t b = n / Lout;
        const int p = (int)(n - b * Lout);
        float* yb = y + b * (int64_t)Cout * Lout + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < Cout) {
                float v = (sub == 0 ? acc0[r] : acc1[r]) + (bias != nullptr ? bias[co] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                yb[(int64_t)co * Lout] = v;
            }
        }
    }
}

}  // namespace

extern "C" int lbm_conv1d_f32(const float* x, const float* w, const float* bias, float* y, int64_t B, int32_t Cin,
                              int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                              lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Lin >= 1, LBM_EINVAL, "lbm_conv1d_f32: bad sizes");
    LBM_REQUIRE(ksize >= 1 && stride >= 1 && pad >= 0, LBM_EINVAL, "lbm_conv1d_f32: bad ksize/stride/pad");
    LBM_REQUIRE(Lin + 2 * pad >= ksize, LBM_EINVAL, "lbm_conv1d_f32: kernel (%d) larger than padded input (%d)", ksize, Lin + 2 * pad);
    if (B == 0) return LBM_OK;
    LBM_REQUIRE(x && w && y, LBM_EINVAL, "lbm_conv1d_f32: NULL pointer");
    const int Lout = (Lin + 2 * pad - ksize) / stride + 1;
    const int64_t Ntot = B * Lout;
    const int64_t gx = (Ntot + BN - 1) / BN;
    const int gy = (Cout + BM - 1) / BM;
    LBM_REQUIRE(gx <= 0x7fffffffLL && gy <= 65535, LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    hipLaunchKernelGGL(conv1d_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), 0, lbm::as_stream(stream), x, w, bias, y,
                       Ntot, Cin, Lin, Cout, Lout, ksize, stride, pad, relu);
    return lbm::check_launch("lbm_conv1d_f32");
}
