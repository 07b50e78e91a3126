// FastICA fixed-point statistics (ICA *fit*): one pass over the whitened recording per
// iteration.  Replaces the body of sklearn's `_ica_par` loop that `mne.preprocessing.ICA(method='fastica')`
// runs (training/preprocess.py:91-98, multi-train/withpsd.py:127, preprocessing_unsupervised.py:39):
//     y = W z;   g = fun(y);   gxt[i][j] = sum_t g_i(t) z_j(t);   gsum[i] = sum_t fun'(y_i(t))
// and, with W = I and the identity "nonlinearity", the second-moment / sum pass of the PCA whitening step.
//
// Shape of the work: z is [n <= 64 rows, T samples] with T ~ 1e6, so both products are GEMMs whose long axis
// is time.  One wavefront owns a run of 32-sample tiles and keeps everything in registers:
//   (1) Y^T tile [32 samples x 32 comps] = Z^T W^T on v_mfma_f32_32x32x2_f32 (A = samples x channels read
//       straight from global, B = W held in registers for the whole kernel);
//   (2) the nonlinearity on the 16 accumulator registers (lane = component, registers = samples);
//   (3) M[channel][comp] += Z tile x G: the accumulator of (1) IS the B operand of (3) -- register r of lane
//       (j, h) holds sample (r&3)+8(r>>2)+4h, which is exactly the k index MFMA number r sums over when the
//       A operand is loaded as Z[channel j][that sample].  No transpose, no LDS.
// HBM-bound by design (z is read once per iteration; the second read of each tile hits L1/L2).
// Partial sums are fp32 over at most 8192 samples per wavefront, then float64 in a fixed order (deterministic).
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int kWaves = 4;
constexpr int kMaxTilesPerWave = 256;

struct Plan {
    int nct;            // 32-wide tiles per side (1: n <= 32, 2: n <= 64)
    int np;             // padded side
    int64_t ntiles;     // 32-sample tiles
    int tpw;            // tiles per wavefront
    int64_t nwg;        // workgroups (= partial blocks)
    size_t stride;      // floats per partial block
};

Plan make_plan(int64_t T, int n) {
    Plan p;
    p.nct = n <= 32 ? 1 : 2;
    p.np = 32 * p.nct;
    p.ntiles = (T + 31) / 32;
    int64_t tpw = (p.ntiles + 4095) / 4096;
    if (tpw < 1) tpw = 1;
    if (tpw > kMaxTilesPerWave) tpw = kMaxTilesPerWave;
    p.tpw = (int)tpw;
    p.nwg = (p.ntiles + (int64_t)kWaves * tpw - 1) / ((int64_t)kWaves * tpw);
    if (p.nwg < 1) p.nwg = 1;
    p.stride = (size_t)p.np * p.np + p.np;
    return p;
}

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int NCT>
__global__ __launch_bounds__(64 * kWaves) void ica_stats_kernel(const float* __restrict__ z, int64_t ldz, int64_t T, int n,
                                                                const float* __restrict__ W, int mode, float alpha,
                                                                float* __restrict__ part, int tpw, int64_t ntiles) {
    constexpr int NP = 32 * NCT;
    constexpr int STRIDE = NP * NP + NP;
    extern __shared__ __align__(16) float s_red[];   // [kWaves][STRIDE]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;

    // B operand of step (1): W^T[channel 2t+h of tile kt][component j of tile ct], zero beyond n
    float wr[NCT][NCT][16];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int kt = 0; kt < NCT; ++kt)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int comp = ct * 32 + j, ch = kt * 32 + 2 * t + h;
                wr[ct][kt][t] = (comp < n && ch < n) ? W[comp * n + ch] : 0.f;
            }

    f32x16 M[NCT][NCT];   // [channel tile][component tile]
    float gs[NCT];
#pragma unroll
    for (int a = 0; a < NCT; ++a) {
        gs[a] = 0.f;
#pragma unroll
        for (int b = 0; b < NCT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) M[a][b][r] = 0.f;
    }

    const int64_t tile0 = ((int64_t)blockIdx.x * kWaves + wave) * tpw;
    const int64_t tile1 = tile0 + tpw < ntiles ? tile0 + tpw : ntiles;
    for (int64_t tile = tile0; tile < tile1; ++tile) {
        const int64_t s0 = tile * 32;
        const int64_t sj = s0 + j < T ? s0 + j : T - 1;
        const bool okj = s0 + j < T;
        float xa[NCT][16], xm[NCT][16];
#pragma unroll
        for (int kt = 0; kt < NCT; ++kt) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                int ch = kt * 32 + 2 * t + h;
                ch = ch < n ? ch : n - 1;
                xa[kt][t] = z[ch * ldz + sj];
            }
            int chj = kt * 32 + j;
            chj = chj < n ? chj : n - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t s = s0 + acc_row(r, h);
                xm[kt][r] = z[chj * ldz + (s < T ? s : T - 1)];
            }
        }
        f32x16 y[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) y[ct][r] = 0.f;
#pragma unroll
            for (int kt = 0; kt < NCT; ++kt)
#pragma unroll
                for (int t = 0; t < 16; ++t)
                    y[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(okj ? xa[kt][t] : 0.f, wr[ct][kt][t], y[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = y[ct][r];
                float g, gp;
                if (mode == LBM_ICA_FUN_LOGCOSH) {   // g = tanh(a y), g' = a (1 - g^2)   (sklearn _logcosh)
                    const float e = __expf(2.f * alpha * v);
                    g = 1.f - 2.f / (e + 1.f);
                    gp = alpha * (1.f - g * g);
                } else if (mode == LBM_ICA_FUN_EXP) {   // g = y exp(-y^2/2), g' = (1 - y^2) exp(-y^2/2)   (_exp)
                    const float e = __expf(-0.5f * v * v);
                    g = v * e;
                    gp = (1.f - v * v) * e;
                } else if (mode == LBM_ICA_FUN_CUBE) {   // g = y^3, g' = 3 y^2   (_cube)
                    g = v * v * v;
                    gp = 3.f * v * v;
                } else {   // identity: second moments and sums
                    g = v;
                    gp = v;
                }
                const bool ok = s0 + acc_row(r, h) < T;
                y[ct][r] = ok ? g : 0.f;
                gs[ct] += ok ? gp : 0.f;
            }
#pragma unroll
        for (int kt = 0; kt < NCT; ++kt)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    M[kt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(xm[kt][r], y[ct][r], M[kt][ct], 0, 0, 0);
    }

    // per-wave block in LDS as [comp][channel] + [comp] sums, then the 4 waves are added in a fixed order
    float* mine = s_red + wave * STRIDE;
#pragma unroll
    for (int kt = 0; kt < NCT; ++kt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(ct * 32 + j) * NP + kt * 32 + acc_row(r, h)] = M[kt][ct][r];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const float tot = gs[ct] + __shfl_xor(gs[ct], 32, 64);
        if (h == 0) mine[NP * NP + ct * 32 + j] = tot;
    }
    __syncthreads();
    float* out = part + (size_t)blockIdx.x * STRIDE;
    for (int i = threadIdx.x; i < STRIDE; i += 64 * kWaves)
        out[i] = (s_red[i] + s_red[STRIDE + i]) + (s_red[2 * STRIDE + i] + s_red[3 * STRIDE + i]);
}

// out[e] = sum over partial blocks, float64, fixed order: lane = element, 16 waves split the blocks, LDS tree
__global__ __launch_bounds__(1024) void ica_reduce_kernel(const float* __restrict__ part, int64_t nwg, int np, int n,
                                                          double* __restrict__ gxt, double* __restrict__ gsum) {
    __shared__ double s_p[16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int stride = np * np + np;
    const int total = n * n + n;
    const int e = blockIdx.x * 64 + lane;
    // element e of the output -> offset in a padded partial block
    int off = 0;
    if (e < n * n) off = (e / n) * np + (e % n);
    else if (e < total) off = np * np + (e - n * n);
    double acc = 0.0;
    if (e < total)
        for (int64_t b = wave; b < nwg; b += 16) acc += (double)part[(size_t)b * stride + off];
    s_p[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && e < total) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += s_p[w][lane];
        if (e < n * n) gxt[e] = t;
        else gsum[e - n * n] = t;
    }
}

}  // namespace

extern "C" size_t lbm_ica_stats_workspace_bytes(int64_t T, int32_t n) {
    if (T <= 0 || n <= 0 || n > 64) return 0;
    const Plan p = make_plan(T, n);
    return (size_t)p.nwg * p.stride * sizeof(float);
}

extern "C" int lbm_ica_stats_f32(const float* z, int64_t ldz, int64_t T, int32_t n, const float* W, int32_t fun, float alpha,
                                 double* gxt, double* gsum, void* workspace, lbm_stream_t stream) {
    LBM_REQUIRE(z && W && gxt && gsum && workspace, LBM_EINVAL, "lbm_ica_stats_f32: NULL pointer");
    LBM_REQUIRE(T > 0 && n > 0 && ldz >= T, LBM_EINVAL, "lbm_ica_stats_f32: bad shape (T=%lld, n=%d, ldz=%lld)", (long long)T, n,
                (long long)ldz);
    LBM_REQUIRE(n <= 64, LBM_EUNSUPPORTED, "lbm_ica_stats_f32: n=%d components (at most 64)", n);
    LBM_REQUIRE(fun >= LBM_ICA_FUN_LOGCOSH && fun <= LBM_ICA_FUN_IDENTITY, LBM_EINVAL, "lbm_ica_stats_f32: unknown fun %d", fun);
    const Plan p = make_plan(T, n);
    LBM_REQUIRE(p.nwg <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_ica_stats_f32: grid too large");
    hipStream_t st = lbm::as_stream(stream);
    float* part = static_cast<float*>(workspace);
    const size_t lds = (size_t)kWaves * p.stride * sizeof(float);
    if (p.nct == 1) {
        hipLaunchKernelGGL(ica_stats_kernel<1>, dim3((unsigned)p.nwg), dim3(64 * kWaves), lds, st, z, ldz, T, n, W, fun, alpha, part,
                           p.tpw, p.ntiles);
    } else {
        auto k = ica_stats_kernel<2>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_ica_stats_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(k, dim3((unsigned)p.nwg), dim3(64 * kWaves), lds, st, z, ldz, T, n, W, fun, alpha, part, p.tpw, p.ntiles);
    }
    int rc = lbm::check_launch("lbm_ica_stats_f32");
    if (rc != LBM_OK) return rc;
    const int total = n * n + n;
    hipLaunchKernelGGL(ica_reduce_kernel, dim3((total + 63) / 64), dim3(1024), 0, st, part, p.nwg, p.np, n, gxt, gsum);
    return lbm::check_launch("lbm_ica_stats_f32(reduce)");
}
