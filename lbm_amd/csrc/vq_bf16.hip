// Codebook distance + argmin with the fp32 products rebuilt from bf16 pieces on the bf16 matrix cores.
//
// Same contract as lbm_vq_argmin_f32 (training/model.py:152-158: d = (|x|^2 + |e|^2) - 2 x.e, first minimum),
// different arithmetic for x.e: every fp32 operand is cut into three bf16 pieces v = p1 + p2 + p3 (each the
// round-to-nearest bf16 of the running residual; the residuals are exact in fp32, and p1+p2+p3 carries >= 24
// significant bits), and the product keeps the six terms down to 2^-18 relative:
//     x.e = sum_k  p3(e)p1(x) + p2(e)p2(x) + p1(e)p3(x) + p2(e)p1(x) + p1(e)p2(x) + p1(e)p1(x)      (fp32 accumulate)
// The dropped terms are < 2^-26 of a product; measured against float64 (tools/probe/bf16x_probe.hip) the result
// is as accurate as the fp32 fmaf chain of the fp32-MFMA kernel (max 2.3 vs 2.3 eps, rms 0.39 vs 0.48 eps of
// sum|x_k e_k|) -- while v_mfma_f32_32x32x16_bf16 retires 16x the multiply-adds per cycle of
// v_mfma_f32_32x32x2_f32, i.e. 16/6 = 2.7x the fp32 matrix rate for fp32-grade results.  Tokens are the same
// except at fp32 near-ties (the rule of SURVEY.md 8(a)), exactly as between two fp32 BLAS libraries.
//
// Layout of the work: workgroup = 128 input rows x a super-tile of 256 codes, 4 waves as 2 row groups x 2 code
// groups, wave tile 64 rows x 128 codes = 8 accumulators of 32x32.  K (<= 256 dims) is consumed 16 at a time:
// per stage the workgroup stages the codes' pre-split pieces (caller-owned table, lbm_vq_split_tables_init) and
// splits its 128 x 16 slice of x on the fly into an LDS image [piece][row][16 bf16] (32-byte rows, 16-byte
// halves swapped on odd row octets so the ds_read_b128 operand reads are conflict-free), double buffered, one
// barrier per stage; 48 MFMAs per wave per stage.
#include "bf16x6.h"

namespace {

// tab: [nst][3][Kp][16] bf16, zero beyond (K, D); the two 16-byte halves of a row are stored in the order of the
// kernel's LDS image (swapped on odd row octets), so a stage's slab is copied to LDS linearly
__global__ __launch_bounds__(256) void vq_split_tables_kernel(const float* __restrict__ codebook, bf16x8* __restrict__ tab, int K,
                                                              int Kp, int D, int nst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (stage, code, half)
    if (i >= (int64_t)nst * Kp * 2) return;
    const int half = (int)(i & 1);
    const int code = (int)((i >> 1) % Kp);
    const int st = (int)((i >> 1) / Kp);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = st * kKC + half * 8 + e;
        v[e] = (code < K && k < D) ? codebook[(int64_t)code * D + k] : 0.f;
    }
    bf16x8 p[3];
    split8(v, p[0], p[1], p[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) tab[(((int64_t)st * 3 + q) * Kp + code) * 2 + (half ^ ((code >> 3) & 1))] = p[q];   // LDS image order
}

// rows[code][piece][256] bf16: the first two pieces of every code over all dims (zero beyond D and K); thread = (code, octet)
__global__ __launch_bounds__(256) void vq_rows_table_kernel(const float* __restrict__ codebook, bf16x8* __restrict__ rows, int K, int Kp,
                                                            int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)Kp * 32) return;
    const int code = (int)(i >> 5), oct = (int)(i & 31);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = oct * 8 + e;
        v[e] = (code < K && d < D) ? codebook[(int64_t)code * D + d] : 0.f;
    }
    bf16x8 p1, p2, p3;
    split8(v, p1, p2, p3);
    rows[(int64_t)code * 64 + oct] = p1;
    rows[(int64_t)code * 64 + 32 + oct] = p2;
}

template <int LAYOUT>
__global__ __launch_bounds__(kThreads, 2) void vq_bf16x6_kernel(const float* __restrict__ x, int64_t ldx, const u32x4* __restrict__ tab,
                                                                const float* __restrict__ e_sqnorm, int64_t* __restrict__ idx_out,
                                                                float* __restrict__ mind, int64_t N, int D, int K, int Kp, int nst,
                                                                int64_t S, int no_xnorm) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* const s_ee = reinterpret_cast<float*>(smem + 2 * kStageBytes);          // [2][kSup]
    float* const s_xx = s_ee + 2 * kSup;                                             // [2][kRows]
    float* const s_bd = s_xx + 2 * kRows;                                            // [2][kRows] best distance per code group
    int* const s_bi = reinterpret_cast<int*>(s_bd + 2 * kRows);                      // [2][kRows]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rg = wave & 1, cg = wave >> 1;
    const int nsup = Kp / kSup;
    const int total = nsup * nst;

    // ---- staging roles
    // x: thread -> (row = tid / 2, octet = tid & 1) of the stage's 128 x 16 slice (8 consecutive lanes write 128
    // contiguous bytes of the LDS image: conflict-free ds_write_b128)
    const int xr = tid >> 1, xo = tid & 1;
    int64_t n = (int64_t)blockIdx.x * kRows + xr;
    n = n < N ? n : N - 1;
    const float* xrow;
    int64_t kstride;
    if constexpr (LAYOUT == LBM_VQ_ROWS) {
        xrow = x + n * ldx;
        kstride = 1;
    } else {
        const int64_t b = n / S, sp = n - b * S;
        xrow = x + b * D * ldx + sp;
        kstride = ldx;
    }
    float xv[8];
    float xxp = 0.f;
    const int ks = (int)kstride;                 // D * ldx < 2^31 is checked by the launcher
    const int tstep = 3 * Kp * 2;                // 16-byte chunks per stage of the table

    // (sup, st) = the stage held in the staging registers
    auto load_stage = [&](int sup, int st) {
        const int kb = st * kKC + xo * 8;
        if (kb + 8 <= D) {
            const float* xp = xrow + (int64_t)kb * ks;
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[e] = xp[e * ks];
        } else {   // ragged last octet: clamped addresses, masked when stored
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb + e < D ? kb + e : D - 1;
                xv[e] = xrow[(int64_t)k * ks];
            }
        }
    };
    // the stage's code slab goes global -> LDS directly (no registers, no ds_write): lane-linear 1 KiB per wave-instruction
    auto copy_codes = [&](int sup, int st, int buf) {
        const u32x4* tp = tab + (st * tstep + sup * (kSup * 2) + tid);
        unsigned char* sb = smem + buf * kStageBytes + tid * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tp + (i >> 1) * (Kp * 2) + (i & 1) * kThreads),
                                             (__attribute__((address_space(3))) void*)(sb + (i >> 1) * kCodePlane + (i & 1) * (kThreads * 16)), 16, 0, 0);
    };
    auto store_stage = [&](int sup, int st, int buf) {
        unsigned char* sb = smem + buf * kStageBytes;
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
        unsigned char* xb = sb + 3 * kCodePlane + swz(xr, xo);
        *reinterpret_cast<bf16x8*>(xb) = p1;
        *reinterpret_cast<bf16x8*>(xb + kXPlane) = p2;
        *reinterpret_cast<bf16x8*>(xb + 2 * kXPlane) = p3;
        if (st == 0) {   // |e|^2 of the super-tile this stage opens
            const int code = sup * kSup + tid;
            s_ee[(sup & 1) * kSup + tid] = code < K ? e_sqnorm[code] : INFINITY;
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

    // all six products of one 16-wide k block for the wave's 2 x 4 tiles, small terms first; the code
    // fragments of tile ct+1 are requested before the MFMAs of tile ct
    const int a_off = swz(cg * 128 + j, h), b_off = 3 * kCodePlane + swz(rg * 64 + j, h);   // swz is linear in 32-row steps
    auto mfma_stage = [&](int buf) {
        const unsigned char* sb = smem + buf * kStageBytes;
        bf16x8 bx[2][3], ae[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) ae[0][p] = *reinterpret_cast<const bf16x8*>(sb + p * kCodePlane + a_off);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int p = 0; p < 3; ++p) bx[rt][p] = *reinterpret_cast<const bf16x8*>(sb + b_off + p * kXPlane + rt * 1024);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            if (ct < 3) {
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    ae[(ct + 1) & 1][p] = *reinterpret_cast<const bf16x8*>(sb + p * kCodePlane + a_off + (ct + 1) * 1024);
            }
            const bf16x8* a_ = ae[ct & 1];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                f32x16 a = acc[rt][ct];
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], bx[rt][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bx[rt][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bx[rt][2], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], bx[rt][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bx[rt][1], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], bx[rt][0], a, 0, 0, 0);
                acc[rt][ct] = a;
            }
        }
    };

    float best[2] = {INFINITY, INFINITY};
    int bidx[2] = {0, 0};
    float xx[2] = {0.f, 0.f};

    int rsup = 0, rst = 0;   // stage in the staging registers
    auto advance = [&]() {
        if (++rst == nst) {
            rst = 0;
            ++rsup;
        }
    };
    load_stage(0, 0);
    copy_codes(0, 0, 0);
    store_stage(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (total > 1) {
        advance();
        load_stage(rsup, rst);
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
                    load_stage(rsup, rst);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_stage(buf);
            if (st == nst - 1) {   // the super-tile is complete: fold its 128 x 64 distances into the running argmin
                if (sup == 0 && !no_xnorm) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const int row = rg * 64 + rt * 32 + j;
                        xx[rt] = s_xx[row] + s_xx[kRows + row];
                    }
                }
                const float* ee = s_ee + (sup & 1) * kSup + cg * 128;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int cl = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float e2 = ee[cl];
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt) {
                            const float dist = (xx[rt] + e2) - 2.f * acc[rt][ct][r];
                            if (dist < best[rt]) {
                                best[rt] = dist;
                                bidx[rt] = sup * kSup + cg * 128 + cl;
                            }
                        }
                    }
                zero_acc();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stage copied by this wave has landed
            __syncthreads();
        }
    }

    // merge the lane halves, then the two code groups (smaller distance, then smaller index)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const float ob = __shfl_xor(best[rt], 32, 64);
        const int oi = __shfl_xor(bidx[rt], 32, 64);
        if (ob < best[rt] || (ob == best[rt] && oi < bidx[rt])) {
            best[rt] = ob;
            bidx[rt] = oi;
        }
        if (h == 0) {
            s_bd[cg * kRows + rg * 64 + rt * 32 + j] = best[rt];
            s_bi[cg * kRows + rg * 64 + rt * 32 + j] = bidx[rt];
        }
    }
    __syncthreads();
    if (tid < kRows) {
        const int64_t row = (int64_t)blockIdx.x * kRows + tid;
        float b0 = s_bd[tid];
        int i0 = s_bi[tid];
        const float b1 = s_bd[kRows + tid];
        const int i1 = s_bi[kRows + tid];
        if (b1 < b0 || (b1 == b0 && i1 < i0)) {
            b0 = b1;
            i0 = i1;
        }
        if (row < N) {
            idx_out[row] = (int64_t)i0;
            if (mind != nullptr) mind[row] = b0;
        }
    }
}

// en[k] >= |e_k|: fp32 sum of squares (relative error <= D 2^-24), square root, 1e-5 of slack; 0 beyond K
__global__ __launch_bounds__(256) void vq_code_norms_kernel(const float* __restrict__ codebook, float* __restrict__ en, int K, int Kp, int D) {
    const int code = blockIdx.x * 256 + threadIdx.x;
    if (code >= Kp) return;
    float s = 0.f;
    if (code < K)
        for (int d = 0; d < D; ++d) {
            const float v = codebook[(int64_t)code * D + d];
            s = fmaf(v, v, s);
        }
    en[code] = sqrtf(s) * 1.00001f;
}

}  // namespace

extern "C" size_t lbm_vq_split_tables_bytes(int32_t K, int32_t D) {
    if (K < 1 || D < 1 || D > 256) return 0;
    return vq_tab_plan(K, D).bytes;
}

extern "C" int lbm_vq_split_tables_init(void* tables, const float* codebook, int32_t K, int32_t D, lbm_stream_t stream) {
    LBM_REQUIRE(tables && codebook, LBM_EINVAL, "lbm_vq_split_tables_init: NULL pointer");
    LBM_REQUIRE(K >= 1 && D >= 1 && D <= 256, LBM_EUNSUPPORTED, "lbm_vq_split_tables_init: K=%d, D=%d (1 <= D <= 256)", K, D);
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_vq_split_tables_init: tables must be 16-byte aligned");
    const VqTabPlan p = vq_tab_plan(K, D);
    const int64_t items = (int64_t)p.nst * p.Kp * 2;
    hipLaunchKernelGGL(vq_split_tables_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, lbm::as_stream(stream), codebook,
                       static_cast<bf16x8*>(tables), K, p.Kp, D, p.nst);
    hipLaunchKernelGGL(vq_code_norms_kernel, dim3((unsigned)((p.Kp + 255) / 256)), dim3(256), 0, lbm::as_stream(stream), codebook,
                       reinterpret_cast<float*>(static_cast<unsigned char*>(tables) + p.piece_bytes), K, p.Kp, D);
    hipLaunchKernelGGL(vq_rows_table_kernel, dim3((unsigned)((p.Kp * 32 + 255) / 256)), dim3(256), 0, lbm::as_stream(stream), codebook,
                       reinterpret_cast<bf16x8*>(static_cast<unsigned char*>(tables) + p.rows_off), K, p.Kp, D);
    return lbm::check_launch("lbm_vq_split_tables_init");
}

// the usage histogram kernel lives in vq.hip
int lbm_vq_hist_launch(const int64_t* idx, unsigned long long* hist, int64_t N, int K, hipStream_t st);

extern "C" int lbm_vq_argmin_bf16x6_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const void* tables,
                                        const float* e_sqnorm, int64_t* idx_out, float* mind_or_null,
                                        unsigned long long* hist_or_null, int64_t N, int32_t D, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0, LBM_EINVAL, "lbm_vq_argmin_bf16x6_f32: negative N");
    LBM_REQUIRE(D >= 1 && D <= 256, LBM_EUNSUPPORTED, "lbm_vq_argmin_bf16x6_f32: D=%d must be in [1,256]", D);
    LBM_REQUIRE(K >= 1, LBM_EINVAL, "lbm_vq_argmin_bf16x6_f32: K must be >= 1");
    LBM_REQUIRE(layout >= 0 && layout <= (LBM_VQ_CHANNEL_MAJOR | LBM_VQ_NO_XNORM), LBM_EINVAL,
                "lbm_vq_argmin_bf16x6_f32: unknown layout %d", layout);
    const int no_xnorm = (layout & LBM_VQ_NO_XNORM) ? 1 : 0;
    layout &= ~LBM_VQ_NO_XNORM;
    if (N == 0) return LBM_OK;
    LBM_REQUIRE(x && tables && e_sqnorm && idx_out, LBM_EINVAL, "lbm_vq_argmin_bf16x6_f32: NULL pointer");
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_vq_argmin_bf16x6_f32: tables must be 16-byte aligned");
    if (layout == LBM_VQ_ROWS)
        LBM_REQUIRE(ldx >= D, LBM_EINVAL, "lbm_vq_argmin_bf16x6_f32: ldx (%lld) < D (%d)", (long long)ldx, D);
    else
        LBM_REQUIRE(S >= 1 && ldx >= S && N % S == 0, LBM_EINVAL,
                    "lbm_vq_argmin_bf16x6_f32: channel-major needs 1 <= S <= ldx and N %% S == 0 (N=%lld, S=%lld, ldx=%lld)",
                    (long long)N, (long long)S, (long long)ldx);
    LBM_REQUIRE(layout == LBM_VQ_ROWS || (int64_t)D * ldx < 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_vq_argmin_bf16x6_f32: D * ldx must be below 2^31");
    const VqTabPlan p = vq_tab_plan(K, D);
    LBM_REQUIRE(p.piece_bytes / 16 < 0x7fffffffULL, LBM_EUNSUPPORTED, "lbm_vq_argmin_bf16x6_f32: table too large");
    const int64_t blocks = (N + kRows - 1) / kRows;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_vq_argmin_bf16x6_f32: grid too large");
    hipStream_t st = lbm::as_stream(stream);
    auto k = layout == LBM_VQ_ROWS ? vq_bf16x6_kernel<LBM_VQ_ROWS> : vq_bf16x6_kernel<LBM_VQ_CHANNEL_MAJOR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_bf16x6_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), kLdsBytes, st, x, ldx, static_cast<const u32x4*>(tables), e_sqnorm,
                       idx_out, mind_or_null, N, D, K, p.Kp, p.nst, S, no_xnorm);
    int rc = lbm::check_launch("lbm_vq_argmin_bf16x6_f32");
    if (rc != LBM_OK) return rc;
    if (hist_or_null != nullptr) {
        LBM_REQUIRE(K <= 32768, LBM_EUNSUPPORTED, "lbm_vq_argmin_bf16x6_f32: histogram supports K <= 32768");
        rc = lbm_vq_hist_launch(idx_out, hist_or_null, N, K, st);
    }
    return rc;
}
