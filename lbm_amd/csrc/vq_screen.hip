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
//       - fp32 accumulation inside and between the 48 MFMAs of a dot product, priced at 16 x 2^-24 (|C| + sum|a b|) per
//         instruction, as if its 16 adds were sequential round-to-nearest ones.  Measured (tools/probe/mfma_acc_probe.hip):
//         <= 4.3 on random operands of mixed magnitude, 7.8 on operands built against an adder that aligns every product
//         to the largest one and drops what falls two bits below its ulp (which is what the matrix core does: the worst
//         case of that model is 9.5)                                                                              4.6e-5
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
// The pieces are those of vq_bf16.hip (bf16x6.h); the codebook comes pre-split in the caller-owned table.
#include "bf16x6.h"
#include <type_traits>

namespace {

constexpr float kAlpha = 1.5e-4f;
constexpr float kBeta = 5e-7f;

// Work layout (that of vq_argmin_dma_kernel, vq.hip): a workgroup = 8 wavefronts = 256 rows, each wavefront keeps ITS 32 rows --
// already cut into two bf16 pieces, 128 VGPRs -- in registers for the whole kernel as the B operand; the codebook streams through
// LDS in tiles of 64 codes as the A operand.  A code's LDS row is its 1 KiB table row [piece 1: 256 bf16 | piece 2: 256 bf16]
// + 16 bytes of padding (ds_read_b128 of lanes = codes is conflict-free at a 260-dword stride), dropped there by LDS-DMA, one
// wave-instruction per code row; two buffers, tile t+1 in flight under tile t's products, one barrier per tile.  The rows are
// split once per kernel (not once per 256-code super-tile as in vq_bf16.hip), and nothing but the code tiles goes through LDS.
constexpr int kSThreads = 512;
constexpr int kSRows = 256;
constexpr int kSTile = 64;
constexpr int kSRowB = 1024 + 16;                       // bytes per code row in LDS
constexpr int kSBufB = kSTile * kSRowB;                 // 66560: one tile of code rows
constexpr int kSConst = 2 * kSBufB;                     // then a ring of three slots of tile constants: ee_lo[64], en[64]
constexpr int kSSlotB = 2 * kSTile * 4;
constexpr int kSLds = kSConst + 3 * kSSlotB;

// cee[k] = ee[k] - beta |ee[k]| (the allowance for the fp32 roundings that form a score), +inf beyond K (padding codes never win)
__global__ __launch_bounds__(256) void vq_screen_consts_kernel(const float* __restrict__ e_sqnorm, float* __restrict__ cee, int K, int Kp) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= Kp) return;
    const float e2 = k < K ? e_sqnorm[k] : INFINITY;
    cee[k] = k < K ? e2 - kBeta * fabsf(e2) : INFINITY;
}

template <int LAYOUT, bool FULLD>
__global__ __launch_bounds__(kSThreads) void vq_screen_kernel(const float* __restrict__ x, int64_t ldx, const unsigned char* __restrict__ rows_tab,
                                                              const float* __restrict__ en_tab, const float* __restrict__ cee,
                                                              const float* __restrict__ e_sqnorm, int64_t* __restrict__ idx_out,
                                                              int64_t* __restrict__ row_list, int* __restrict__ row_count, int64_t N, int D,
                                                              int K, int vec_ok, int64_t S) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * kSRows + wave * 32 + j;
    const int64_t row_c = row < N ? row : N - 1;
    const int ntiles = (K + kSTile - 1) / kSTile;

    // tile -> LDS: wave w copies code rows 8w .. 8w+7; waves 0 and 1 also the tile's 64 + 64 constants
    auto copy_tile = [&](int tile, int buf) {
        unsigned char* dst = smem + buf * kSBufB;
        unsigned char* cst = smem + kSConst + (tile % 3) * kSSlotB;
        const unsigned char* src = rows_tab + ((int64_t)tile * kSTile + wave * 8) * 1024 + lane * 16;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + (wave * 8 + r) * kSRowB), 16, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cee + tile * kSTile + lane),
                                             (__attribute__((address_space(3))) void*)(cst), 4, 0, 0);
        if (wave == 1)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(en_tab + tile * kSTile + lane),
                                             (__attribute__((address_space(3))) void*)(cst + kSTile * 4), 4, 0, 0);
    };
    copy_tile(0, 0);

    // ---- this lane's half of its row (k-step s covers dims 16 s .. 16 s + 15; lane half h holds 16 s + 8 h .. + 7): all 128
    // loads go out at once (one exposed round trip per workgroup), each step's values are cut into bf16 pieces as they arrive
    bf16x8 x1[16], x2[16];
    float xx = 0.f;
    {
        const float* xr = nullptr;      // LBM_VQ_ROWS: this lane's row
        const float* xblk = nullptr;    // channel-major: workgroup-uniform base + 32-bit lane offset (the launcher checks the span)
        unsigned xoff = 0;
        int ld = 1;
        if constexpr (LAYOUT == LBM_VQ_ROWS) {
            xr = x + row_c * ldx + 8 * h;
        } else {
            const int64_t b0 = ((int64_t)blockIdx.x * kSRows) / S;
            const int64_t b = row_c / S, sp = row_c - b * S;
            ld = (int)ldx;
            xblk = x + b0 * D * ldx;
            xoff = (unsigned)((b - b0) * D * ldx + sp) + (unsigned)(8 * h * ld);
        }
        constexpr int NB = (LAYOUT == LBM_VQ_ROWS && !FULLD) ? 2 : 1;   // (the masked scalar row loads need their registers: two batches)
        constexpr int SPB = 16 / NB;
#pragma unroll
        for (int half = 0; half < NB; ++half) {
            float t[SPB][8];
#pragma unroll
            for (int u = 0; u < SPB; ++u) {
                const int s_ = half * SPB + u;
                if constexpr (LAYOUT == LBM_VQ_ROWS) {
                    if (FULLD && vec_ok) {
                        const float4 v0 = *reinterpret_cast<const float4*>(xr + 16 * s_), v1 = *reinterpret_cast<const float4*>(xr + 16 * s_ + 4);
                        t[u][0] = v0.x; t[u][1] = v0.y; t[u][2] = v0.z; t[u][3] = v0.w;
                        t[u][4] = v1.x; t[u][5] = v1.y; t[u][6] = v1.z; t[u][7] = v1.w;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int d = 16 * s_ + 8 * h + e;
                            t[u][e] = xr[16 * s_ + ((FULLD || d < D) ? e : -(16 * s_ + 8 * h))];   // clamped to the row's first value
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int d = 16 * s_ + 8 * h + e;
                        const float* col = xblk + (int64_t)((16 * s_ + e) * ld);     // uniform part of the address
                        t[u][e] = (FULLD || d < D) ? col[xoff] : xblk[xoff - (unsigned)(8 * h * ld)];
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // every request is out before the first value is touched
#pragma unroll
            for (int u = 0; u < SPB; ++u) {
                const int s_ = half * SPB + u;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    asm volatile("" : "+v"(t[u][e]));
                    v[e] = (FULLD || 16 * s_ + 8 * h + e < D) ? t[u][e] : 0.f;
                    xx = fmaf(v[e], v[e], xx);
                }
                bf16x8 p3;
                split8(v, x1[s_], x2[s_], p3);
            }
        }
    }
    xx += __shfl_xor(xx, 32, 64);
    const float ax = kAlpha * sqrtf(xx) * 1.00001f;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float best = INFINITY, second = INFINITY;
    int bidx = 0;
    // lower bound of one score -> running (best, its code, second best).  The row constant |x|^2 is left out of the screen's
    // scores (a common offset of a row's bounds; the rounding it adds to the fp32 chain's score is inside beta).
    auto fold = [&](float a, float e2, float nk, int code) {
        const float lo = fmaf(-ax, nk, fmaf(-2.f, a, e2));
        second = __builtin_amdgcn_fmed3f(best, second, lo);
        bidx = lo < best ? code : bidx;
        best = fminf(best, lo);
    };
    // the same with the winner tracked as a small literal (the k-step) inside a half-tile: no index arithmetic per score
    int loc = 0;
    auto fold_loc = [&](float a, float e2, float nk, int step) {
        const float lo = fmaf(-ax, nk, fmaf(-2.f, a, e2));
        second = __builtin_amdgcn_fmed3f(best, second, lo);
        loc = lo < best ? step : loc;
        best = fminf(best, lo);
    };
    // A tile is searched as two half-tiles of 32 codes, each one accumulator chain of 16 k-steps x three piece products (small
    // terms first; a single chain of v_mfma_f32_32x32x16_bf16 runs at the full rate); the code fragments of step s+1 are
    // requested before the products of step s.  The 16 scores per lane of the PREVIOUS half-tile are folded into the running
    // minimum between this half-tile's MFMAs, one score per k-step: in a phase of its own that work would leave the matrix pipe
    // idle (every wave of the workgroup reaches it at the same time), in the gaps of the MFMA stream it is nearly free.  Its
    // constants come from a ring of three slots, so the DMA of tile t+1 never touches what tile t-1's scores still need.
    auto do_half = [&](const int tile, auto half_c, f32x16& cur, const f32x16& prev, auto has_prev) {
        constexpr int half = decltype(half_c)::value;
        constexpr bool fold_prev = decltype(has_prev)::value;
        const int buf = tile & 1;
        if (half == 0 && tile + 1 < ntiles) copy_tile(tile + 1, buf ^ 1);   // that buffer's last readers passed the barrier below
        const unsigned char* ap = smem + buf * kSBufB + (half * 32 + j) * kSRowB + h * 16;
        // the previous half-tile: codes 32-63 of tile - 1 (slot (tile + 2) % 3) or codes 0-31 of this tile
        const int ptile = half == 0 ? tile - 1 : tile;
        const float* pe = reinterpret_cast<const float*>(smem + kSConst + ((ptile + 3) % 3) * kSSlotB) + (1 - half) * 32 + 4 * h;
        const int pcode = ptile * kSTile + (1 - half) * 32 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] = 0.f;
        const float best_in = best;   // (fold_prev) the previous half-tile's winner, if any, becomes a code index after the steps
        loc = 0;
        bf16x8 af[2][2];   // [parity][piece 1, piece 2]
        float ke[2], kn[2];
        auto load_s = [&](int s_, int par) {
            af[par][0] = *reinterpret_cast<const bf16x8*>(ap + 32 * s_);
            af[par][1] = *reinterpret_cast<const bf16x8*>(ap + 512 + 32 * s_);
            if constexpr (fold_prev) {   // step s folds accumulator register s = code (s & 3) + 8 (s >> 2) (+ 4 h) of that half
                ke[par] = pe[(s_ & 3) + 8 * (s_ >> 2)];
                kn[par] = pe[kSTile + (s_ & 3) + 8 * (s_ >> 2)];
            }
        };
        load_s(0, 0);
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            if (s_ + 1 < 16) load_s(s_ + 1, (s_ + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const bf16x8* f = af[s_ & 1];
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], x1[s_], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], x2[s_], cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], x1[s_], cur, 0, 0, 0);
            if constexpr (fold_prev) {
                fold_loc(prev[s_], ke[s_ & 1], kn[s_ & 1], s_);
#pragma unroll
                for (int i = 0; i < 3; ++i) {   // one MFMA, then two of the six fold instructions, ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (fold_prev) bidx = best < best_in ? pcode + (loc & 3) + 8 * (loc >> 2) : bidx;
        if (half == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows this wave copied for the next tile have landed
            __syncthreads();
        }
    };
    f32x16 accA, accB;
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    do_half(0, C0{}, accA, accB, std::false_type{});
    do_half(0, C1{}, accB, accA, std::true_type{});
    for (int tile = 1; tile < ntiles; ++tile) {
        do_half(tile, C0{}, accA, accB, std::true_type{});
        do_half(tile, C1{}, accB, accA, std::true_type{});
    }
    {   // the last half-tile's scores
        const float* pe = reinterpret_cast<const float*>(smem + kSConst + ((ntiles - 1) % 3) * kSSlotB) + 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 e4 = *reinterpret_cast<const float4*>(pe + 8 * q), n4 = *reinterpret_cast<const float4*>(pe + kSTile + 8 * q);
            const float e2[4] = {e4.x, e4.y, e4.z, e4.w}, nk[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(accB[4 * q + u], e2[u], nk[u], (ntiles - 1) * kSTile + 32 + 4 * h + 8 * q + u);
        }
    }

    // merge the lane halves: (best, its code, second best) of the union
    {
        const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
        const int oi = __shfl_xor(bidx, 32, 64);
        second = fminf(fmaxf(best, ob), fminf(second, os));
        if (ob < best || (ob == best && oi < bidx)) {
            best = ob;
            bidx = oi;
        }
    }
    if (h == 0 && row < N) {
        // upper bound of the best code's fp32 score, in the same offset frame as the lower bounds (the row constant beta |x|^2 was
        // left out of every lower bound, so it enters twice here)
        const float e2 = e_sqnorm[bidx];
        const float bb = ax * en_tab[bidx] + kBeta * (2.f * xx + fabsf(e2));
        const float up = best + 2.f * bb * 1.00001f;
        idx_out[row] = (int64_t)bidx;
        if (!(second > up)) {   // another code may reach the fp32 minimum (NaN rows land here too): full fp32 search
            const int pos = atomicAdd(row_count, 1);
            row_list[pos] = row;
        }
    }
}

}  // namespace

// vq.hip
int lbm_vq_hist_launch(const int64_t* idx, unsigned long long* hist, int64_t N, int K, hipStream_t st);
int lbm_vq_refine_launch(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                         int64_t* idx_out, const int64_t* row_list, const int* row_count, int64_t N, int D, int K, hipStream_t st);

extern "C" size_t lbm_vq_screen_workspace_bytes(int64_t N, int32_t K) {
    if (N < 0 || K < 1) return 0;
    return 256 + (size_t)vq_tab_plan(K, 256).Kp * sizeof(float) + (size_t)N * sizeof(int64_t);
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
    if (layout != LBM_VQ_ROWS) {   // floats spanned by the rows of one workgroup (32-bit lane offsets in the screening kernel)
        const int64_t span = ((int64_t)kSRows / S + 2) * D * ldx;
        LBM_REQUIRE(span < (1LL << 30), LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: a block of %d rows spans %lld floats (limit 2^30)", kSRows,
                    (long long)span);
    }
    const VqTabPlan p = vq_tab_plan(K, D);
    const int64_t blocks = (N + kSRows - 1) / kSRows;
    hipStream_t st = lbm::as_stream(stream);
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    int* row_count = reinterpret_cast<int*>(ws);
    float* cee = reinterpret_cast<float*>(ws + 256);
    int64_t* row_list = reinterpret_cast<int64_t*>(ws + 256 + (size_t)p.Kp * sizeof(float));
    hipError_t e = hipMemsetAsync(row_count, 0, 256, st);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32: hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(vq_screen_consts_kernel, dim3((unsigned)((p.Kp + 255) / 256)), dim3(256), 0, st, e_sqnorm, cee, K, p.Kp);
    const unsigned char* tb = static_cast<const unsigned char*>(tables);
    const float* en_tab = reinterpret_cast<const float*>(tb + p.piece_bytes);
    const int vec_ok = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0) ? 1 : 0;
    auto k = layout == LBM_VQ_ROWS ? (D == 256 ? vq_screen_kernel<LBM_VQ_ROWS, true> : vq_screen_kernel<LBM_VQ_ROWS, false>)
                                   : (D == 256 ? vq_screen_kernel<LBM_VQ_CHANNEL_MAJOR, true> : vq_screen_kernel<LBM_VQ_CHANNEL_MAJOR, false>);
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, kSLds);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kSThreads), kSLds, st, x, ldx, tb + p.rows_off, en_tab, cee, e_sqnorm, idx_out, row_list,
                       row_count, N, D, K, vec_ok, S);
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
