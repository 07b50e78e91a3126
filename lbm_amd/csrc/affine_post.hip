// Fused ICA affine map + epoch post-processing (a3 + a4/a5/a6 in one pass over the epoch).
//
//   y[e,o,t]  = sum_c A[o,c] x[e,c,t] + b[o]                    (ica.apply, training/preprocess.py:144)
//   base      = mean(y[:n_baseline]);  y_base = y - base        (mne.Epochs baseline=(None,0), :262-271)
//   keep[e]   = 0 if any channel has max - min > ptp_thresh     (reject=dict(eeg=150e-6), multi-train/withpsd.py:213-219)
//   y_std     = (y_base - mean_t) / (std_t + eps)               (z-score, training/preprocess.py:279-287)
//   row_scale = std_t + eps   per (epoch, channel)              (lets the Welch kernel read y_std: PSD(y_base) = row_scale^2 PSD(y_std))
//
// The separate kernels read and write the epoch twice (affine r+w, post r+w = 720 000 B per 30x1500 epoch); this one reads it once
// and writes it once (360 000 B), and the un-standardised epochs never exist unless the caller asks for them.
//
// One workgroup = one epoch x 32 output channels, NW wavefronts, each owning NT tiles of 32 time samples.  The channel mix is an
// fp32 MFMA product with TIME as the M operand: v_mfma_f32_32x32x2_f32, A operand = x[c = 2s + h][t0 + j] loaded straight from
// global memory (lanes 0..31 read 128 contiguous bytes of one channel row, lanes 32..63 of the next), B operand = A^T held in
// registers for the whole kernel.  The accumulator layout then gives every lane 16 time samples of ONE output channel per tile, so
// the row statistics (sum, baseline sum, min, max, sum of squared deviations) are plain in-lane reductions over NT*16 values, one
// shuffle between the lane halves and one tiny LDS exchange between the waves -- the whole epoch stays in accumulator registers
// between the two statistics passes, and the outputs leave as 16-byte stores.  fp32 MFMA is an exact ascending-k fmaf chain.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kMaxWaves = 8;

// KCH: input channels are consumed in KCH blocks of 32 (Cin <= 32 KCH); NT: 32-sample tiles per wavefront
template <int KCH, int NT, bool VEC>
__global__ __launch_bounds__(64 * kMaxWaves) void affine_post_kernel(const float* __restrict__ x, const float* __restrict__ A,
                                                                     const float* __restrict__ bias, float* __restrict__ y_base,
                                                                     float* __restrict__ y_std, float* __restrict__ row_scale,
                                                                     uint8_t* __restrict__ keep, int Cin, int Cout, int T, int nb,
                                                                     float thresh, int do_std, float eps) {
    __shared__ float s_part[2][kMaxWaves][32][4];
    // (the wave number through readfirstlane: hipcc then knows that everything derived from it is wave-uniform -- tile-in-range tests
    // become scalar branches instead of per-lane compares and selects)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t e = blockIdx.x;
    const int o = blockIdx.y * 32 + j;          // this lane's output channel
    const bool o_ok = o < Cout;
    const int tw0 = wave * NT * 32;              // first time sample of this wave

    // k-step q multiplies the channel pair (2q, 2q + 1): lane half h reads row c0c(q) + h with c0c = min(2q, Cin - 2) a WAVE-UNIFORM
    // row base, so every load is "scalar base + per-lane offset" and always inside the epoch.  Where the clamp makes a step re-read
    // a row an earlier step already covered (ragged K: 2q + 1 >= Cin), the weight of that lane is zero instead -- no branch and no
    // select in the load / MFMA stream.  (A non-finite sample then reaches a 0 * x product: NaN in, NaN out, as in numpy.)
    const int cmax2 = Cin >= 2 ? Cin - 2 : 0;
    const int hrow = Cin >= 2 ? h : 0;
    // B operand: A^T.  bw[q] = A[o][row read by (q, h)] if that row is new at step q, else 0
    float bw[KCH * 16];
#pragma unroll
    for (int q = 0; q < KCH * 16; ++q) {
        const int c0c = 2 * q < cmax2 ? 2 * q : cmax2;
        const int cr = c0c + hrow;
        const int prev_hi = (2 * q - 2 < cmax2 ? 2 * q - 2 : cmax2) + 1;      // highest row read by step q - 1
        const bool fresh = (q == 0 || cr > prev_hi) && (Cin >= 2 || h == 0);
        const float a = A[(int64_t)(o_ok ? o : 0) * Cin + cr];
        bw[q] = (fresh && o_ok) ? a : 0.f;
    }
    const float bo = (bias != nullptr && o_ok) ? bias[o] : 0.f;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = bo;       // the bias rides as the accumulator's start value

    // A operand: lane (j, h) of k-step q reads x[e][c0c(q) + h][t0 + j]; samples past T are clamped to the row's last one (they land
    // in output rows that are never used).  Offsets inside the epoch are 32-bit (Cin * T < 2^31 is checked by the host).
    // The lane part of an address is an UNSIGNED 32-bit byte offset next to a scalar row base: the load's own "scalar base + 32-bit lane
    // offset" form, no 64-bit per-lane add per load (Cin * T * 4 < 2^32: host check).
    const float* xe = x + e * (int64_t)Cin * T;
    unsigned toff[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = tw0 + 32 * i + j;
        toff[i] = 4u * (unsigned)(hrow * T + (t < T ? t : T - 1));
    }
#ifndef LBM_AP_G
#define LBM_AP_G 4
#endif
    constexpr int G = LBM_AP_G;                   // k-steps per load group: G*NT loads in flight per lane, double buffered
    constexpr int NG = KCH * 16 / G;
    float xv[2][G][NT];
    auto load_group = [&](int g, float (&dst)[G][NT]) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int q = g * G + u;
            const float* xr = xe + (int64_t)(2 * q < cmax2 ? 2 * q : cmax2) * T;   // scalar
#pragma unroll
#if defined(LBM_AP_ABLATE) && (LBM_AP_ABLATE & 4)        // timing experiment: no loads
            for (int i = 0; i < NT; ++i) dst[u][i] = __int_as_float((int)(size_t)xr + (int)toff[i]) * 1e-30f;
#else
            for (int i = 0; i < NT; ++i) dst[u][i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(xr) + toff[i]);
#endif
        }
    };
    load_group(0, xv[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) load_group(g + 1, xv[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);       // the next group's loads are issued BEFORE this group's MFMA stream ...
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[g & 1][u][i], bw[g * G + u], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);       // ... and are not sunk below it
    }

    // accumulator register r of tile i = time tw0 + 32 i + 8 (r / 4) + 4 h + (r % 4), channel o.  Whether a tile lies inside the row (and
    // inside the baseline) as a whole, not at all, or in part is wave-uniform: only the one ragged tile of a row pays for per-sample
    // selects (round 3: the selects and compares of every tile were 700 of the kernel's 2 200 vector instructions per wave).  Sums
    // run in the same order with or without the selects: same bits.
    const int tl0 = tw0 + 4 * h;
    float vsum = 0.f, bsum = 0.f, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ts = tw0 + 32 * i;             // first sample of the tile (uniform)
        if (ts + 32 <= T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][r];
                vsum += v;
                vmin = fminf(vmin, v);
                vmax = fmaxf(vmax, v);
            }
        } else if (ts < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tl0 + 32 * i + 8 * (r >> 2) + (r & 3);
                const float v = acc[i][r];
                const bool ok = t < T;
                vsum += ok ? v : 0.f;
                vmin = fminf(vmin, ok ? v : INFINITY);
                vmax = fmaxf(vmax, ok ? v : -INFINITY);
            }
        }
        if (ts + 32 <= nb) {                     // nb <= T
#pragma unroll
            for (int r = 0; r < 16; ++r) bsum += acc[i][r];
        } else if (ts < nb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tl0 + 32 * i + 8 * (r >> 2) + (r & 3);
                bsum += (t < nb) ? acc[i][r] : 0.f;
            }
        }
    }
    vsum += __shfl_xor(vsum, 32, 64);
    bsum += __shfl_xor(bsum, 32, 64);
    vmin = fminf(vmin, __shfl_xor(vmin, 32, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, 32, 64));
    if (h == 0) {
        s_part[0][wave][j][0] = vsum;
        s_part[0][wave][j][1] = bsum;
        s_part[0][wave][j][2] = vmin;
        s_part[0][wave][j][3] = vmax;
    }
    __syncthreads();
    vsum = 0.f; bsum = 0.f; vmin = INFINITY; vmax = -INFINITY;
    for (int w = 0; w < nw; ++w) {               // fixed order: deterministic
        const float4 p = *reinterpret_cast<const float4*>(&s_part[0][w][j][0]);
        vsum += p.x; bsum += p.y; vmin = fminf(vmin, p.z); vmax = fmaxf(vmax, p.w);
    }
    const float base = nb > 0 ? bsum / (float)nb : 0.f;
    const float mean = vsum / (float)T - base;   // mean of the baseline-corrected row
    if (keep != nullptr && wave == 0 && h == 0 && o_ok && (vmax - vmin) > thresh) keep[e] = 0;

    float ss = 0.f;
    const float shift = base + mean;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int ts = tw0 + 32 * i;
        if (ts + 32 <= T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d = acc[i][r] - shift;
                ss = fmaf(d, d, ss);
            }
        } else if (ts < T) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = tl0 + 32 * i + 8 * (r >> 2) + (r & 3);
                const float d = acc[i][r] - shift;
                ss = fmaf(t < T ? d : 0.f, d, ss);
            }
        }
    }
    float denom = 1.f;
    if (do_std) {
        ss += __shfl_xor(ss, 32, 64);
        if (h == 0) s_part[1][wave][j][0] = ss;
        __syncthreads();
        ss = 0.f;
        for (int w = 0; w < nw; ++w) ss += s_part[1][w][j][0];
        denom = sqrtf(ss / (float)T) + eps;
        if (row_scale != nullptr && wave == 0 && h == 0 && o_ok) row_scale[e * (int64_t)Cout + o] = denom;
    }
    if (!o_ok) return;
    const float inv = 1.f / denom;               // one reciprocal per row; the product differs from a division by <= 1 ulp
    const int64_t rowoff = (e * (int64_t)Cout + o) * T;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            const int t = tl0 + 32 * i + 8 * rq;
            if (t >= T) continue;
            float vb[4], vs[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vb[u] = acc[i][4 * rq + u] - base;
                vs[u] = (acc[i][4 * rq + u] - shift) * inv;
            }
            if constexpr (VEC) {                 // T % 4 == 0 and 16-byte aligned outputs: a quad is in range as a whole
#if defined(LBM_AP_ABLATE) && (LBM_AP_ABLATE & 1)      // timing experiment (results garbage): the same stores, 1 KiB per wave-instruction
                if (do_std) *reinterpret_cast<float4*>(y_std + e * (int64_t)Cout * T + ((wave * NT + i) * 4 + rq) * 256 + lane * 4) = make_float4(vs[0], vs[1], vs[2], vs[3]);
                continue;
#elif defined(LBM_AP_ABLATE) && (LBM_AP_ABLATE & 2)    // timing experiment: no stores
                if (vs[0] == 1.2345f) y_std[0] = vb[0];
                continue;
#endif
                if (y_base != nullptr) *reinterpret_cast<float4*>(y_base + rowoff + t) = make_float4(vb[0], vb[1], vb[2], vb[3]);
                if (do_std) *reinterpret_cast<float4*>(y_std + rowoff + t) = make_float4(vs[0], vs[1], vs[2], vs[3]);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (t + u < T) {
                        if (y_base != nullptr) y_base[rowoff + t + u] = vb[u];
                        if (do_std) y_std[rowoff + t + u] = vs[u];
                    }
            }
        }
}

template <int KCH, int NT>
int launch(const float* x, const float* A, const float* b, float* y_base, float* y_std, float* row_scale, uint8_t* keep, int64_t E,
           int Cin, int Cout, int T, int nb, float thresh, int do_std, float eps, hipStream_t st) {
    const int nw = (T + 32 * NT - 1) / (32 * NT);
    const dim3 grid((unsigned)E, (unsigned)((Cout + 31) / 32));
    const bool vec = (T % 4 == 0) && ((((uintptr_t)y_base) | ((uintptr_t)y_std)) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((affine_post_kernel<KCH, NT, true>), grid, dim3(64 * nw), 0, st, x, A, b, y_base, y_std, row_scale, keep, Cin,
                           Cout, T, nb, thresh, do_std, eps);
    else
        hipLaunchKernelGGL((affine_post_kernel<KCH, NT, false>), grid, dim3(64 * nw), 0, st, x, A, b, y_base, y_std, row_scale, keep, Cin,
                           Cout, T, nb, thresh, do_std, eps);
    return lbm::check_launch("lbm_affine_post_f32");
}

}  // namespace

extern "C" int lbm_affine_post_supported(int32_t Cin, int32_t Cout, int32_t T) {
    return (Cin >= 1 && Cin <= 64 && Cout >= 1 && Cout <= 256 && T >= 1 && T <= 32 * 6 * kMaxWaves) ? 1 : 0;
}

extern "C" int lbm_affine_post_f32(const float* x, const float* A, const float* b, float* y_base, float* y_std, float* row_scale,
                                   uint8_t* keep, int64_t E, int32_t Cin, int32_t Cout, int32_t T, int32_t n_baseline,
                                   float ptp_thresh, int32_t standardize_mode, float eps, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && Cin >= 1 && Cout >= 1 && T >= 1, LBM_EINVAL, "lbm_affine_post_f32: bad sizes");
    LBM_REQUIRE(n_baseline >= 0 && n_baseline <= T, LBM_EINVAL, "lbm_affine_post_f32: n_baseline must be in [0, T]");
    LBM_REQUIRE(standardize_mode == LBM_STD_NONE || standardize_mode == LBM_STD_EPOCH_CHANNEL, LBM_EINVAL,
                "lbm_affine_post_f32: unknown standardize_mode %d", standardize_mode);
    LBM_REQUIRE(lbm_affine_post_supported(Cin, Cout, T), LBM_EUNSUPPORTED,
                "lbm_affine_post_f32: supports Cin <= 64, Cout <= 256, T <= 1536 (got %d, %d, %d): use lbm_affine_ch_f32 + lbm_epoch_post_f32",
                Cin, Cout, T);
    if (E == 0) return LBM_OK;
    const int do_std = (standardize_mode == LBM_STD_EPOCH_CHANNEL && y_std != nullptr) ? 1 : 0;
    LBM_REQUIRE(x && A, LBM_EINVAL, "lbm_affine_post_f32: NULL pointer");
    LBM_REQUIRE(E <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_affine_post_f32: grid too large");
    LBM_REQUIRE(!(y_base == nullptr && !do_std && keep == nullptr && row_scale == nullptr), LBM_EINVAL, "lbm_affine_post_f32: no output requested");
    hipStream_t st = lbm::as_stream(stream);
    if (keep != nullptr) {
        hipError_t err = hipMemsetAsync(keep, 1, (size_t)E, st);
        if (err != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_affine_post_f32: memset keep: %s", hipGetErrorString(err));
    }
    // tiles per wave: the smallest NT that covers T with <= 8 waves (more, shorter waves spread the MFMA work over the SIMDs)
#define LBM_AP_LAUNCH(KCH, NT) \
    return launch<KCH, NT>(x, A, b, y_base, y_std, row_scale, keep, E, Cin, Cout, T, n_baseline, ptp_thresh, do_std, eps, st)
    if (Cin <= 32) {
        if (T <= 32 * 4 * kMaxWaves) LBM_AP_LAUNCH(1, 4);
        LBM_AP_LAUNCH(1, 6);
    }
    if (T <= 32 * 4 * kMaxWaves) LBM_AP_LAUNCH(2, 4);
    LBM_AP_LAUNCH(2, 6);
#undef LBM_AP_LAUNCH
}
