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
#include <atomic>

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

// LDS-DMA through inline assembly.  With __builtin_amdgcn_global_load_lds anywhere in a loop the compiler (ROCm 7.2) stops counting
// its LDS waits: every s_waitcnt before an MFMA becomes lgkmcnt(0), which also waits for the fragments just requested for the NEXT
// k-step, so the prefetch hides nothing (measured: 58 cycles per MFMA from a wave alone on its SIMD, 32 being the pipe's rate).
// Issued this way the copy is invisible to that bookkeeping and the waits come out counted (lgkmcnt(2)).  The wave base address goes
// to M0 (saved and restored: the compiler keeps its own idea of M0); completion is the caller's s_waitcnt vmcnt(0) + barrier, as before.
__device__ __forceinline__ void lds_dma16(const void* src, const void* lds_dst) {
    const unsigned dst = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}
// the same with a wave-uniform source base (SGPR pair) and a 32-bit lane offset: no address registers per piece
__device__ __forceinline__ void lds_dma16_s(const void* src_uniform, unsigned lane_off, const void* lds_dst) {
    const unsigned dst = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(src_uniform), "s"(dst) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* src, const void* lds_dst) {
    const unsigned dst = (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)lds_dst;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
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
    const unsigned lane16 = lane * 16;
    auto copy_tile = [&](int tile, int buf) {
        unsigned char* dst = smem + buf * kSBufB;
        unsigned char* cst = smem + kSConst + (tile % 3) * kSSlotB;
        const unsigned char* src = rows_tab + ((int64_t)tile * kSTile + wave * 8) * 1024;
#pragma unroll
        for (int r = 0; r < 8; ++r) lds_dma16_s(src + r * 1024, lane16, dst + (wave * 8 + r) * kSRowB);
        if (wave == 0) lds_dma4(cee + tile * kSTile + lane, cst);
        if (wave == 1) lds_dma4(en_tab + tile * kSTile + lane, cst + kSTile * 4);
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

// ------------------------------------------------------------------------------------------ staggered, persistent form (D = 256)
// vq_screen_kernel spends a quarter of a K = 512 launch outside the tile loop: every workgroup loads and splits its 256 rows, merges
// and writes, is torn down and relaunched, with nothing else on the CU to overlap it (the 131 KB of tile buffers and the 204
// registers per lane allow no second workgroup).  Here the workgroup is persistent and its two GROUPS of four waves walk 128-row
// blocks half a period apart: the code tiles stream through LDS cyclically (iteration i holds tile i mod ntiles -- the minimum over
// the codes does not care where a block enters the cycle), a block takes ntiles + 2 iterations of its group -- rows in flight, rows
// split, then ntiles tiles -- and while one group is in its two memory iterations the other one has the matrix pipe to itself.
// Every wave takes part in every barrier, whatever its group is doing, and the iteration count is one number for the whole
// workgroup.  The rows of a block are requested straight into the registers the pieces live in (one 128-dword array per lane:
// raw floats until the split rewrites it in place).
//
// What the s_memtime stamps of tools/probe/stamps_pipe.py say about one period (K = 512: 10 iterations, 87k cycles, 49k of them
// matrix work): an iteration with both groups on tiles takes 7.9k cycles for 6.1k of MFMA; a group alone on its SIMDs 5.3k for 3.1k
// (a chain of dependent MFMAs plus the tile copy's issue slots, with no second wave to fill the gaps); and the iteration in which
// a group requests its rows 12k, because 128 KB arrive at some 13 B/cycle per CU while the whole chip streams (an L2 touch-ahead
// brought the request itself to 7.9k and cost the other iterations as much; a second iteration for the rows to land gained
// nothing either: the request holds the CU's memory pipe, and the other group's tile copies queue behind it).
template <int LAYOUT>
__global__ __launch_bounds__(kSThreads) void vq_screen_pipe_kernel(const float* __restrict__ x, int64_t ldx, const unsigned char* __restrict__ rows_tab,
                                                                   const float* __restrict__ en_tab, const float* __restrict__ cee,
                                                                   const float* __restrict__ e_sqnorm, int64_t* __restrict__ idx_out,
                                                                   int64_t* __restrict__ row_list, int* __restrict__ row_count, int64_t N,
                                                                   int K, int vec_ok, int64_t S) {
    constexpr int D = 256, GR = 128;   // rows per group block
    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int g = wave >> 2, wg = wave & 3;
    const int ntiles = (K + kSTile - 1) / kSTile;
    const int P = ntiles + 2, off = g * (P / 2);
    const int nblk = (int)((N + GR - 1) / GR);
    const int stride = 2 * (int)gridDim.x;
    auto blocks_of = [&](int gg) {
        const int fb = 2 * (int)blockIdx.x + gg;
        return fb < nblk ? (nblk - 1 - fb) / stride + 1 : 0;
    };
    const int nb0 = blocks_of(0), nb1 = blocks_of(1), nbg = g ? nb1 : nb0;
    const int it0 = nb0 * P, it1 = nb1 ? (P / 2) + nb1 * P : 0;
    const int I = it0 > it1 ? it0 : it1;   // iterations of the whole workgroup

    // one code row (1 KiB piece) of a tile: this wave's r-th of its 8
    const unsigned lane16 = lane * 16;
    auto copy_row = [&](int tile, int buf, int r) {
        lds_dma16_s(rows_tab + ((int64_t)tile * kSTile + wave * 8 + r) * 1024, lane16, smem + buf * kSBufB + (wave * 8 + r) * kSRowB);
    };
    auto copy_consts = [&](int tile, int slot) {
        unsigned char* cst = smem + kSConst + slot * kSSlotB;
        if (wave == 0) lds_dma4(cee + tile * kSTile + lane, cst);
        if (wave == 1) lds_dma4(en_tab + tile * kSTile + lane, cst + kSTile * 4);
    };
    auto copy_tile = [&](int tile, int buf, int slot) {
#pragma unroll
        for (int r = 0; r < 8; ++r) copy_row(tile, buf, r);
        copy_consts(tile, slot);
    };

    // this lane's half of its row: dims 16 s + 8 h + e in component e & 3 of reg[2 s + e / 4] (raw float bits); after the split reg[2 s]
    // is piece 1 of k-step s and reg[2 s + 1] piece 2
    u32x4v reg[32];        // reg[2 s], reg[2 s + 1]: the two 16-byte halves of k-step s (4-register tuples, as the MFMA wants them)
    int64_t row = 0;       // the row of the block in the registers
    auto request_rows = [&](int blk) {
        const int64_t r0 = (int64_t)blk * GR;
        row = r0 + wg * 32 + j;
        const int64_t row_c = row < N ? row : N - 1;
        if constexpr (LAYOUT == LBM_VQ_ROWS) {
            const float* xr = x + row_c * ldx + 8 * h;
            if (vec_ok) {
#pragma unroll
                for (int s_ = 0; s_ < 16; ++s_) {
                    const u32x4v v0 = *reinterpret_cast<const u32x4v*>(xr + 16 * s_), v1 = *reinterpret_cast<const u32x4v*>(xr + 16 * s_ + 4);
                    reg[2 * s_] = v0;
                    reg[2 * s_ + 1] = v1;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 128; ++i) reg[i >> 2][i & 3] = __float_as_uint(xr[16 * (i >> 3) + (i & 7)]);
            }
        } else {
            const int64_t b0 = r0 / S;                        // uniform: the block's first epoch
            const int64_t b = row_c / S, sp = row_c - b * S;
            const int ld = (int)ldx;
            const float* xblk = x + b0 * D * ldx;             // scalar base + 32-bit lane offset (the launcher checks the span)
            const unsigned xoff = (unsigned)((b - b0) * D * ldx + sp) + (unsigned)(8 * h * ld);
#pragma unroll
            for (int i = 0; i < 128; ++i) {
                const float* col = xblk + (int64_t)((16 * (i >> 3) + (i & 7)) * ld);
                reg[i >> 2][i & 3] = __float_as_uint(col[xoff]);
            }
        }
    };
    float xx = 0.f, ax = 0.f, best = INFINITY, second = INFINITY;
    int bidx = 0, loc = 0;
    auto split_rows = [&]() {
        xx = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] = __uint_as_float(reg[2 * s_ + (e >> 2)][e & 3]);
                xx = fmaf(v[e], v[e], xx);
            }
            bf16x8 p1, p2, p3;
            split8(v, p1, p2, p3);
            reg[2 * s_] = __builtin_bit_cast(u32x4v, p1);
            reg[2 * s_ + 1] = __builtin_bit_cast(u32x4v, p2);
        }
        xx += __shfl_xor(xx, 32, 64);
        ax = kAlpha * sqrtf(xx) * 1.00001f;
        best = INFINITY;
        second = INFINITY;
        bidx = 0;
    };
    auto X1 = [&](int s_) { return __builtin_bit_cast(bf16x8, reg[2 * s_]); };
    auto X2 = [&](int s_) { return __builtin_bit_cast(bf16x8, reg[2 * s_ + 1]); };

    auto fold = [&](float a, float e2, float nk, int code) {
        const float lo = fmaf(-ax, nk, fmaf(-2.f, a, e2));
        second = __builtin_amdgcn_fmed3f(best, second, lo);
        bidx = lo < best ? code : bidx;
        best = fminf(best, lo);
    };
    auto fold_loc = [&](float a, float e2, float nk, int step) {
        const float lo = fmaf(-ax, nk, fmaf(-2.f, a, e2));
        second = __builtin_amdgcn_fmed3f(best, second, lo);
        loc = lo < best ? step : loc;
        best = fminf(best, lo);
    };
    // One tile (64 codes, two half-tiles of 32) from buffer `buf`: 32 k-steps of three MFMAs.  The scores of the half-tile before
    // (the previous tile's second half into accB, this tile's first into accA) are folded between the MFMAs -- see vq_screen_kernel.
    // Code fragments and constants are requested kPF k-steps ahead, straight across the half-tile boundary: one step (96 cycles of
    // matrix pipe) is less than an LDS read takes to come back, and a wave alone on its SIMD then waits on every step.
    f32x16 accA, accB;
    auto do_tile = [&](int buf, auto has_prev, int slot, int pslot, int tile, int ptile, bool cp, int ntile, int nslot) {
        constexpr bool first = !decltype(has_prev)::value;
        constexpr int kPF = 1, kRing = kPF + 1;
        const unsigned char* ap = smem + buf * kSBufB + j * kSRowB + h * 16;
        const float* pe0 = reinterpret_cast<const float*>(smem + kSConst + pslot * kSSlotB) + 32 + 4 * h;   // previous tile, second half
        const float* pe1 = reinterpret_cast<const float*>(smem + kSConst + slot * kSSlotB) + 4 * h;          // this tile, first half
        bf16x8 af[kRing][2];
        float ke[kRing], kn[kRing];
        auto load_t = [&](int t) {
            const int st = t % kRing, hf = t >> 4, s_ = t & 15;
            af[st][0] = *reinterpret_cast<const bf16x8*>(ap + hf * 32 * kSRowB + 32 * s_);
            af[st][1] = *reinterpret_cast<const bf16x8*>(ap + hf * 32 * kSRowB + 512 + 32 * s_);
            if (hf == 1 || !first) {
                const float* pe = hf ? pe1 : pe0;
                ke[st] = pe[(s_ & 3) + 8 * (s_ >> 2)];
                kn[st] = pe[kSTile + (s_ & 3) + 8 * (s_ >> 2)];
            }
        };
#pragma unroll
        for (int t = 0; t < kPF; ++t) load_t(t);
        float best_in = best;
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            const int st = t % kRing, hf = t >> 4, s_ = t & 15;
            const bool folds = hf == 1 || !first;
            f32x16& cur = hf ? accB : accA;
            const f32x16& prev = hf ? accA : accB;
            if (s_ == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) cur[r] = 0.f;
                best_in = best;
                loc = 0;
            }
            if (t + kPF < 32) load_t(t + kPF);
            __builtin_amdgcn_sched_barrier(0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st][1], X1(s_), cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st][0], X2(s_), cur, 0, 0, 0);
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st][0], X1(s_), cur, 0, 0, 0);
            if (folds) {
                fold_loc(prev[s_], ke[st], kn[st], s_);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // the next tile's copy, a piece every other k-step of the first half: a piece costs its wave ~60 cycles of issue between
            // MFMAs and 160 in a burst of eight at the top of the iteration, where both waves of a SIMD would be issuing theirs
            if (cp && (t & 1) && t < 16) copy_row(ntile, buf ^ 1, t >> 1);
            if (cp && t == 16) copy_consts(ntile, nslot);
            if (s_ == 15 && folds) {
                const int pcode = (hf ? tile * kSTile : ptile * kSTile + 32) + 4 * h;
                bidx = best < best_in ? pcode + (loc & 3) + 8 * (loc >> 2) : bidx;
            }
        }
    };

    if (I == 0) return;   // uniform
    copy_tile(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int tile = 0, slot = 0;            // tile and ring slot of iteration i
    int ph = -off, kblk = 0;           // this group's position: phase ph (0: rows in flight, 1: split, 2 ..: tiles) of its block kblk
    for (int i = 0; i < I; ++i) {
        const int buf = i & 1;
        const int ntile = tile + 1 == ntiles ? 0 : tile + 1, nslot = slot == 2 ? 0 : slot + 1, pslot = slot == 0 ? 2 : slot - 1;
        const bool cp = i + 1 < I;                          // that buffer's (and slot's) last readers passed the previous barrier
        const bool active = ph >= 0 && kblk < nbg;          // uniform per wave
        // waves in a tile iteration copy between their MFMAs, those that split copy afterwards
        if (cp && !(active && ph >= 1)) copy_tile(ntile, buf ^ 1, nslot);
        // The s_waitcnt at the head of the split and tile paths is for the compiler's bookkeeping only (nothing of this wave's is in
        // flight there, the last iteration closed on vmcnt(0)): without it the compiler takes the row registers to be still loading on
        // these paths and counts vmcnt down between the MFMAs and through the split -- on the tile copies, as it turns out.
        if (active && ph == 0) request_rows(2 * (int)blockIdx.x + g + kblk * stride);   // they land by this iteration's closing wait
        else if (active && ph == 1) {
            __builtin_amdgcn_s_waitcnt(0x0f70);
            split_rows();
            if (cp) copy_tile(ntile, buf ^ 1, nslot);
        } else if (active && ph >= 2) {
            __builtin_amdgcn_s_waitcnt(0x0f70);

            const int ptile = tile == 0 ? ntiles - 1 : tile - 1;
            if (ph == 2) do_tile(buf, std::false_type{}, slot, pslot, tile, ptile, cp, ntile, nslot);
            else do_tile(buf, std::true_type{}, slot, pslot, tile, ptile, cp, ntile, nslot);
            if (ph == P - 1) {   // the block's last tile: its second half-tile's scores, the merge of the lane halves, the verdict
                const float* pe = reinterpret_cast<const float*>(smem + kSConst + slot * kSSlotB) + 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 e4 = *reinterpret_cast<const float4*>(pe + 8 * q), n4 = *reinterpret_cast<const float4*>(pe + kSTile + 8 * q);
                    const float e2[4] = {e4.x, e4.y, e4.z, e4.w}, nk[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) fold(accB[4 * q + u], e2[u], nk[u], tile * kSTile + 32 + 4 * h + 8 * q + u);
                }
                const float ob = __shfl_xor(best, 32, 64), os = __shfl_xor(second, 32, 64);
                const int oi = __shfl_xor(bidx, 32, 64);
                second = fminf(fmaxf(best, ob), fminf(second, os));
                if (ob < best || (ob == best && oi < bidx)) {
                    best = ob;
                    bidx = oi;
                }
                if (h == 0 && row < N) {
                    const float e2 = e_sqnorm[bidx];
                    const float bb = ax * en_tab[bidx] + kBeta * (2.f * xx + fabsf(e2));
                    const float up = best + 2.f * bb * 1.00001f;
                    idx_out[row] = (int64_t)bidx;
                    if (!(second > up)) {
                        const int pos = atomicAdd(row_count, 1);
                        row_list[pos] = row;
                    }
                }
            }
        }
        // The next tile and a block's requested rows have landed (the copies are invisible to the compiler, so the wait is spelled out).
        // Giving the rows a second iteration to land (closing the first on a counted wait) was measured and gains nothing: 128 KB per
        // group arrive at some 13 B/cycle while every CU streams, the request holds the CU's memory pipe for 10k cycles either way, and
        // the other group's tile copies queue behind it.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (++ph == P) {   // advance this group's clock
            ph = 0;
            ++kblk;
        }
        tile = ntile;
        slot = nslot;
    }
}

}  // namespace

// vq.hip
int lbm_vq_hist_launch(const int64_t* idx, unsigned long long* hist, int64_t N, int K, hipStream_t st);
int lbm_vq_refine_launch(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                         int64_t* idx_out, const int64_t* row_list, const int* row_count, int64_t N, int D, int K, hipStream_t st);

namespace {
// The opt-in to more than 64 KiB of dynamic LDS is a per-function, per-device attribute: set once, not on every call.
int dyn_lds_once(const void* fn, int slot) {
    static std::atomic<bool> done[4][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && done[slot][dev].load(std::memory_order_acquire)) return LBM_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kSLds);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (dev >= 0) done[slot][dev].store(true, std::memory_order_release);
    return LBM_OK;
}
}  // namespace

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
    LBM_REQUIRE(hist_or_null == nullptr || K <= 32768, LBM_EUNSUPPORTED, "lbm_vq_argmin_screened_f32: histogram supports K <= 32768");
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
    int rc;
    if (D == 256) {   // persistent, two staggered groups per workgroup
        auto kp = layout == LBM_VQ_ROWS ? vq_screen_pipe_kernel<LBM_VQ_ROWS> : vq_screen_pipe_kernel<LBM_VQ_CHANNEL_MAJOR>;
        rc = dyn_lds_once(reinterpret_cast<const void*>(kp), layout == LBM_VQ_ROWS ? 0 : 1);
        if (rc != LBM_OK) return rc;
        const int64_t nblk = (N + 127) / 128;
        int64_t grid = (nblk + 1) / 2;
        if (grid > 256) grid = 256;
        hipLaunchKernelGGL(kp, dim3((unsigned)grid), dim3(kSThreads), kSLds, st, x, ldx, tb + p.rows_off, en_tab, cee, e_sqnorm, idx_out,
                           row_list, row_count, N, K, vec_ok, S);
        rc = lbm::check_launch("lbm_vq_argmin_screened_f32(screen)");
    } else {
        auto k = layout == LBM_VQ_ROWS ? vq_screen_kernel<LBM_VQ_ROWS, false> : vq_screen_kernel<LBM_VQ_CHANNEL_MAJOR, false>;
        rc = dyn_lds_once(reinterpret_cast<const void*>(k), layout == LBM_VQ_ROWS ? 2 : 3);
        if (rc != LBM_OK) return rc;
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kSThreads), kSLds, st, x, ldx, tb + p.rows_off, en_tab, cee, e_sqnorm, idx_out,
                           row_list, row_count, N, D, K, vec_ok, S);
        rc = lbm::check_launch("lbm_vq_argmin_screened_f32(screen)");
    }
    if (rc != LBM_OK) return rc;
    rc = lbm_vq_refine_launch(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, row_list, row_count, N, D, K, st);
    if (rc != LBM_OK) return rc;
    if (hist_or_null != nullptr) rc = lbm_vq_hist_launch(idx_out, hist_or_null, N, K, st);   // K was checked with the other arguments
    return rc;
}
