// Codebook argmin with the fp32 result and a fraction of its matrix time: screen on the bf16 matrix cores, refine in fp32.
//
// Contract: idx[n] = exactly what lbm_vq_argmin_f32 returns (training/model.py:152-158 evaluated by the fp32 fmaf chains of
// vq.hip, first minimum wins) -- not "close to it".  How:
//
//  1. SCREEN (this file).  Scores s~[n,k] = ee[k] - 2 x_n.e_k with x.e from THREE bf16 piece products (x1 e2 + x2 e1 + x1 e1,
//     pieces as in vq_bf16.hip, fp32 accumulate) on v_mfma_f32_32x32x16_bf16: 3/16 of the fp32 matrix time.  Its error
//     against the fp32 chain's score s[n,k] is bounded per (row, code):
//         |s~ - s| <= B[n,k] = alpha |x_n| |e_k| + beta (|x_n|^2 + |ee[k]|)
//       - dropped piece products: |x e - (x1 e1 + x1 e2 + x2 e1)| <= 3.01 * 2^-18 |x e| per term (bf16 = 8 significant bits,
//         round to nearest: |v - v1| <= 2^-9 |v|, |v - v1 - v2| <= 2^-18 |v|)                                    1.15e-5
//       - fp32 accumulation inside and between the 48 MFMAs of a dot product, taken as fully sequential round-to-nearest
//         adds (16 per instruction, <= 2^-20 of sum|terms| each; the hardware's adder tree is at least that good)  4.6e-5
//       - the fp32 chain being reproduced: 256 fmaf, <= 256 * 2^-24 sum|x e|                                     1.53e-5
//       x2 (the score doubles the dot product), Cauchy-Schwarz sum|x_d e_kd| <= |x||e_k|, 3 % of slack for the rounded
//       norms: alpha = 1.5e-4.  beta = 5e-7 covers the two or three fp32 roundings that form a score on either side.
//     A lane keeps, per row, the smallest LOWER bound lo = s~ - B with its code, and the second smallest.  If the second
//     smallest lower bound exceeds the UPPER bound of the best code (lo + 2B), no other code can reach the fp32 minimum and
//     the best code IS lbm_vq_argmin_f32's answer (strictly: ties are impossible then).  Otherwise the row is appended to a
//     list.  Exact duplicates of a code row always tie and always go to the list.
//  2. REFINE (vq.hip, vq_argmin_kernel<.., LIST>).  The listed rows -- 1-2 % on the shipped model's codebooks -- are searched
//     again over the whole codebook by the fp32 kernel's own code.
//
// Workgroup layout, staging and the LDS image are vq_bf16.hip's (128 rows x 256-code super-tiles, k in stages of 16, codes'
// pieces by LDS-DMA from the pre-split table, x split on the fly); two of the three piece planes are used.
#include "bf16x6.h"

namespace {

constexpr float kAlpha = 1.5e-4f;
constexpr float kBeta = 5e-7f;
// LDS: the two stage buffers, then ee_lo[2][kSup], en[2][kSup], xx[2][kRows], and the merge arrays best/second/idx [2][kRows]
constexpr int kStage2 = 2 * kCodePlane + 2 * kXPlane;   // two piece planes of the codes, two of x: 24 KB per stage
constexpr int kScreenLds = 2 * kStage2 + 2 * 2 * kSup * 4 + 2 * kRows * 4 + 3 * 2 * kRows * 4;

template <int LAYOUT>
__global__ __launch_bounds__(kThreads, 2) void vq_screen_kernel(const float* __restrict__ x, int64_t ldx, const u32x4* __restrict__ tab,
                                                                const float* __restrict__ en_tab, const float* __restrict__ e_sqnorm,
                                                                int64_t* __restrict__ idx_out, int64_t* __restrict__ row_list,
                                                                int* __restrict__ row_count, int64_t N, int D, int K, int Kp, int nst,
                                                                int64_t S, int no_xnorm) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* const s_ee = reinterpret_cast<float*>(smem + 2 * kStage2);          // [2][kSup]  ee[k] - beta |ee[k]|
    float* const s_en = s_ee + 2 * kSup;                                             // [2][kSup]  >= |e_k|
    float* const s_xx = s_en + 2 * kSup;                                             // [2][kRows]
    float* const s_bd = s_xx + 2 * kRows;                                            // [2][kRows] best lower bound per code group
    float* const s_bs = s_bd + 2 * kRows;                                            // [2][kRows] second best
    int* const s_bi = reinterpret_cast<int*>(s_bs + 2 * kRows);                      // [2][kRows]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rg = wave & 1, cg = wave >> 1;
    const int nsup = Kp / kSup;
    const int total = nsup * nst;

    // ---- staging roles (as vq_bf16x6_kernel): thread -> (row = tid / 2, octet = tid & 1) of the stage's 128 x 16 slice of x
    const int xr = tid >> 1, xo = tid & 1;
    int64_t n = (int64_t)blockIdx.x * kRows + xr;
    n = n < N ? n : N - 1;
    // x is addressed as a workgroup-uniform base (scalar registers) + a 32-bit lane offset: one address register per load
    // instead of two (this kernel has none to spare; the launcher checks that a workgroup's rows span < 2^30 floats)
    const float* xblk;
    unsigned xoff;
    int ks;
    if constexpr (LAYOUT == LBM_VQ_ROWS) {
        const int64_t n0 = (int64_t)blockIdx.x * kRows;
        xblk = x + n0 * ldx;
        xoff = (unsigned)((n - n0) * ldx);
        ks = 1;
    } else {
        const int64_t b0 = ((int64_t)blockIdx.x * kRows) / S;
        const int64_t b = n / S, sp = n - b * S;
        xblk = x + b0 * D * ldx;
        xoff = (unsigned)((b - b0) * D * ldx + sp);
        ks = (int)ldx;
    }
    float xv[8];
    float xxp = 0.f;
    const int tstep = 3 * Kp * 2;                // 16-byte chunks per stage of the table

    auto load_stage = [&](int st) {
        const int kb = st * kKC + xo * 8;
        if (kb + 8 <= D) {
            const unsigned o = xoff + (unsigned)(kb * ks);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = xblk[o + (unsigned)(e * ks)];
        } else {   // ragged last octet: clamped addresses, masked when stored
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb + e < D ? kb + e : D - 1;
                xv[e] = xblk[xoff + (unsigned)(k * ks)];
            }
        }
    };
    // the stage's first two piece planes go global -> LDS directly: lane-linear 1 KiB per wave-instruction
    auto copy_codes = [&](int sup, int st, int buf) {
        const u32x4* tp = tab + (st * tstep + sup * (kSup * 2) + tid);
        unsigned char* sb = smem + buf * kStage2 + tid * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tp + (i >> 1) * (Kp * 2) + (i & 1) * kThreads),
                                             (__attribute__((address_space(3))) void*)(sb + (i >> 1) * kCodePlane + (i & 1) * (kThreads * 16)), 16, 0, 0);
    };
    auto store_stage = [&](int sup, int st, int buf) {
        unsigned char* sb = smem + buf * kStage2;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            asm volatile("" : "+v"(xv[e]));
            v[e] = xv[e];
        }
        const int kb = st * kKC + xo * 8;
        if (kb + 8 > D) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = kb + e < D ? v[e] : 0.f;
        }
        if (sup == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) xxp = fmaf(v[e], v[e], xxp);
            if (st == nst - 1) s_xx[xo * kRows + xr] = xxp;
        }
        bf16x8 p1, p2, p3;
        split8(v, p1, p2, p3);
        unsigned char* xb = sb + 2 * kCodePlane + swz(xr, xo);
        *reinterpret_cast<bf16x8*>(xb) = p1;
        *reinterpret_cast<bf16x8*>(xb + kXPlane) = p2;
        if (st == 0) {   // |e|^2 (less its rounding allowance) and the norm bound of the super-tile this stage opens
            const int code = sup * kSup + tid;
            const float e2 = code < K ? e_sqnorm[code] : INFINITY;
            s_ee[(sup & 1) * kSup + tid] = code < K ? e2 - kBeta * fabsf(e2) : INFINITY;
            s_en[(sup & 1) * kSup + tid] = en_tab[code];
        }
    };

    f32x16 acc[2][4];   // [row tile][code tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    // the three products of one 16-wide k block for the wave's 2 x 4 tiles, small terms first
    const int a_off = swz(cg * 128 + j, h), b_off = 2 * kCodePlane + swz(rg * 64 + j, h);   // swz is linear in 32-row steps
    auto mfma_stage = [&](int buf) {
        const unsigned char* sb = smem + buf * kStage2;
        bf16x8 bx[2][2], ae[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p) ae[0][p] = *reinterpret_cast<const bf16x8*>(sb + p * kCodePlane + a_off);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int p = 0; p < 2; ++p) bx[rt][p] = *reinterpret_cast<const bf16x8*>(sb + b_off + p * kXPlane + rt * 1024);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct < 3) {
#pragma unroll
                for (int p = 0; p < 2; ++p)
                    ae[(ct + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(sb + p * kCodePlane + a_off + (ct + 1) * 1024);
            }
            const bf16x8* a_ = ae[ct & 1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16 a = acc[rt][ct];
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bx[rt][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bx[rt][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bx[rt][0], a, 0, 0, 0);
                acc[rt][ct] = a;
            }
        }
    };

    float best[2] = {INFINITY, INFINITY}, second[2] = {INFINITY, INFINITY};
    int bidx[2] = {0, 0};
    float ax[2] = {0.f, 0.f};

    int rsup = 0, rst = 0;   // stage in the staging registers
    auto advance = [&]() {
        if (++rst == nst) {
            rst = 0;
            ++rsup;
        }
    };
    load_stage(0);
    copy_codes(0, 0, 0);
    store_stage(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (total > 1) {
        advance();
        load_stage(rst);
    }
    int g = 0;
    for (int sup = 0; sup < nsup; ++sup) {
        for (int st = 0; st < nst; ++st, ++g) {
            const int buf = g & 1;
            if (g + 1 < total) {   // stage g+1 (in registers since the previous iteration) -> LDS; request stage g+2
                store_stage(rsup, rst, buf ^ 1);
                copy_codes(rsup, rst, buf ^ 1);
                if (g + 2 < total) {
                    advance();
                    load_stage(rst);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_stage(buf);
            if (st == nst - 1) {   // the super-tile is complete: fold its lower bounds into the running (best, second best)
                if (sup == 0) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const int row = rg * 64 + rt * 32 + j;
                        ax[rt] = kAlpha * sqrtf(s_xx[row] + s_xx[kRows + row]) * 1.00001f;
                    }
                }
                // The row constant |x|^2 is left out of the screen's scores (a common offset of a row's bounds; the rounding it
                // adds to the fp32 chain's score is inside beta).  The winning code is tracked as a compile-time number inside
                // the super-tile and turned into a code index once per super-tile.
                const float* ee = s_ee + (sup & 1) * kSup + cg * 128;
                const float* en = s_en + (sup & 1) * kSup + cg * 128;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {   // accumulator registers 4q .. 4q+3 = codes cl0 .. cl0+3
                        const int cl0 = ct * 32 + 8 * q + 4 * h;
                        const float4 e4 = *reinterpret_cast<const float4*>(ee + cl0), n4 = *reinterpret_cast<const float4*>(en + cl0);
                        const float e2[4] = {e4.x, e4.y, e4.z, e4.w}, nk[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int rt = 0; rt < 2; ++rt) {
                                const float sc = fmaf(-2.f, acc[rt][ct][4 * q + u], e2[u]);
                                const float lo = fmaf(-ax[rt], nk[u], sc);
                                second[rt] = __builtin_amdgcn_fmed3f(best[rt], second[rt], lo);
                                bidx[rt] = lo < best[rt] ? sup * kSup + cg * 128 + cl0 + u : bidx[rt];
                                best[rt] = fminf(best[rt], lo);
                                acc[rt][ct][4 * q + u] = 0.f;
                            }
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage copied by this wave has landed
            __syncthreads();
        }
    }

    // merge the lane halves, then the two code groups: (best, its code, second best) of the union
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const float ob = __shfl_xor(best[rt], 32, 64), os = __shfl_xor(second[rt], 32, 64);
        const int oi = __shfl_xor(bidx[rt], 32, 64);
        const float hi2 = fmaxf(best[rt], ob);
        second[rt] = fminf(hi2, fminf(second[rt], os));
        if (ob < best[rt] || (ob == best[rt] && oi < bidx[rt])) {
            best[rt] = ob;
            bidx[rt] = oi;
        }
        if (h == 0) {
            s_bd[cg * kRows + rg * 64 + rt * 32 + j] = best[rt];
            s_bs[cg * kRows + rg * 64 + rt * 32 + j] = second[rt];
            s_bi[cg * kRows + rg * 64 + rt * 32 + j] = bidx[rt];
        }
    }
    __syncthreads();
    if (tid < kRows) {
        const int64_t row = (int64_t)blockIdx.x * kRows + tid;
        float b0 = s_bd[tid], s0 = s_bs[tid];
        int i0 = s_bi[tid];
        const float b1 = s_bd[kRows + tid], s1 = s_bs[kRows + tid];
        const int i1 = s_bi[kRows + tid];
        s0 = fminf(fmaxf(b0, b1), fminf(s0, s1));
        if (b1 < b0 || (b1 == b0 && i1 < i0)) {
            b0 = b1;
            i0 = i1;
        }
        if (row < N) {
            // upper bound of the best code's fp32 score, in the same offset frame as the lower bounds (the row constant
            // beta |x|^2 was left out of every lower bound, so it enters twice here)
            const float xx = s_xx[tid] + s_xx[kRows + tid];
            const float e2 = e_sqnorm[i0];
            const float bb = kAlpha * sqrtf(xx) * 1.00001f * en_tab[i0] + kBeta * (2.f * xx + fabsf(e2));
            const float up = b0 + 2.f * bb * 1.00001f;
            idx_out[row] = (int64_t)i0;
            if (!(s0 > up)) {   // another code may reach the fp32 minimum (NaN rows land here too): full fp32 search
                const int pos = atomicAdd(row_count, 1);
                row_list[pos] = row;
            }
        }
    }
}

}  // namespace

// vq.hip
int lbm_vq_hist_launch(const int64_t* idx, unsigned long long* hist, int64_t N, int K, hipStream_t st);
int lbm_vq_refine_launch(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                         int64_t* idx_out, const int64_t* row_list, const int* row_count, int64_t N, int D, int K, hipStream_t st);

extern "C" size_t lbm_vq_screen_workspace_bytes(int64_t N) {
    if (N < 0) return 0;
    return 256 + (size_t)N * sizeof(int64_t);
}

extern "C" int lbm_vq_argmin_screened_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const float* codebook, const void* tables,
                                          const float* e_sqnorm, int64_t* idx_out, unsigned long long* hist_or_null, void* workspace,
                                          int64_t N, int32_t D, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0, LBM_EINVAL, "lbm_vq_argmin_screened_f32: negative N");
    LBM_REQUIRE(D >= 8 && D <= 256 && (D & 7) == 0, LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: D=%d must be a multiple of 8 in [8,256]", D);
    LBM_REQUIRE(K >= 1, LBM_EINVAL, "lbm_vq_argmin_screened_f32: K must be >= 1");
    LBM_REQUIRE(layout >= 0 && layout <= (LBM_VQ_CHANNEL_MAJOR | LBM_VQ_NO_XNORM), LBM_EINVAL,
                "lbm_vq_argmin_screened_f32: unknown layout %d", layout);
    const int no_xnorm = (layout & LBM_VQ_NO_XNORM) ? 1 : 0;
    layout &= ~LBM_VQ_NO_XNORM;
    if (N == 0) return LBM_OK;
    LBM_REQUIRE(x && codebook && tables && e_sqnorm && idx_out && workspace, LBM_EINVAL, "lbm_vq_argmin_screened_f32: NULL pointer");
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0 && ((uintptr_t)codebook & 15) == 0, LBM_EINVAL,
                "lbm_vq_argmin_screened_f32: tables and codebook must be 16-byte aligned");
    LBM_REQUIRE(((uintptr_t)workspace & 7) == 0, LBM_EINVAL, "lbm_vq_argmin_screened_f32: workspace must be 8-byte aligned");
    if (layout == LBM_VQ_ROWS)
        LBM_REQUIRE(ldx >= D, LBM_EINVAL, "lbm_vq_argmin_screened_f32: ldx (%lld) < D (%d)", (long long)ldx, D);
    else
        LBM_REQUIRE(S >= 1 && ldx >= S && N % S == 0, LBM_EINVAL,
                    "lbm_vq_argmin_screened_f32: channel-major needs 1 <= S <= ldx and N %% S == 0 (N=%lld, S=%lld, ldx=%lld)",
                    (long long)N, (long long)S, (long long)ldx);
    LBM_REQUIRE(layout == LBM_VQ_ROWS || (int64_t)D * ldx < 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_vq_argmin_screened_f32: D * ldx must be below 2^31");
    LBM_REQUIRE(N < 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: N must be below 2^31 (row counter)");
    {   // floats spanned by the rows of one workgroup (32-bit lane offsets in the screening kernel)
        const int64_t span = layout == LBM_VQ_ROWS ? (int64_t)kRows * ldx : ((int64_t)kRows / S + 2) * D * ldx;
        LBM_REQUIRE(span < (1LL << 30), LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: a block of %d rows spans %lld floats (limit 2^30)", kRows,
                    (long long)span);
    }
    const VqTabPlan p = vq_tab_plan(K, D);
    LBM_REQUIRE(p.piece_bytes / 16 < 0x7fffffffULL, LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: table too large");
    const int64_t blocks = (N + kRows - 1) / kRows;
    hipStream_t st = lbm::as_stream(stream);
    int* row_count = static_cast<int*>(workspace);
    int64_t* row_list = reinterpret_cast<int64_t*>(static_cast<unsigned char*>(workspace) + 256);
    hipError_t e = hipMemsetAsync(row_count, 0, 256, st);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    auto k = layout == LBM_VQ_ROWS ? vq_screen_kernel<LBM_VQ_ROWS> : vq_screen_kernel<LBM_VQ_CHANNEL_MAJOR>;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kScreenLds);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    const float* en_tab = reinterpret_cast<const float*>(static_cast<const unsigned char*>(tables) + p.piece_bytes);
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), kScreenLds, st, x, ldx, static_cast<const u32x4*>(tables), en_tab, e_sqnorm,
                       idx_out, row_list, row_count, N, D, K, p.Kp, p.nst, S, no_xnorm);
    int rc = lbm::check_launch("lbm_vq_argmin_screened_f32(screen)");
    if (rc != LBM_OK) return rc;
    rc = lbm_vq_refine_launch(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, row_list, row_count, N, D, K, st);
    if (rc != LBM_OK) return rc;
    if (hist_or_null != nullptr) {
        LBM_REQUIRE(K <= 32768, LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: histogram supports K <= 32768");
        rc = lbm_vq_hist_launch(idx_out, hist_or_null, N, K, st);
    }
    return rc;
}
