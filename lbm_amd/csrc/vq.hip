// VQ-VAE nearest-codebook search (a12/a13): fp32 MFMA distance + fused running argmin.
//
// d[n,k] = (|x_n|^2 + |e_k|^2) - 2 x_n.e_k, idx[n] = first k attaining the minimum
// (training/model.py:90-95, 152-158).  The [N,K] distance matrix never leaves registers.
//
// Mapping (v_mfma_f32_32x32x2_f32, exact fp32 fmaf chains):
//   A operand = codebook tile (rows i = 32 codes), B operand = x^T (columns j = 32 input rows).
//   A wavefront owns 32 input rows for the whole kernel and keeps them as B fragments in registers
//   (D/2 VGPRs: lane (j, h) holds x[row j][h*D/2 + s] for MFMA step s), so x is read from HBM once.
//   The codebook streams through LDS in tiles of 64 codes ([64][D+4] floats: the +4 pad makes the
//   per-lane ds_read_b128 of 4 consecutive k-steps conflict-free), shared by the 4 waves of the block.
//   In the 32x32 accumulator a lane holds ONE input row (column j = lane & 31) and 16 codes
//   ((r&3) + 8*(r>>2) + 4*(lane>>5)), so the running (min, argmin) is a per-lane register update in
//   ascending code order with strict '<' (first minimum wins); the two lane halves are merged once at
//   the end with an index tie-break, which reproduces torch.argmin's first-minimum rule exactly.
#include <atomic>

#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int kTileCodes = 64;
constexpr int kRowsPerBlock = 128;

// LIST = true is the refinement pass of lbm_vq_argmin_screened_f32: the rows to search are row_list[0 .. *row_count) (any order),
// walked by a fixed grid; everything else -- fragments, fmaf chains, argmin rule -- is the same code, so a refined row gets
// exactly the token the full fp32 search gives it.
template <int DP, int LAYOUT, bool LIST = false>
__global__ __launch_bounds__(kThreads, 2) void vq_argmin_kernel(const float* __restrict__ x, int64_t ldx,
                                                                const float* __restrict__ codebook,
                                                                const float* __restrict__ e_sqnorm,
                                                                int64_t* __restrict__ idx_out, float* __restrict__ mind,
                                                                int64_t N, int D, int K, int vec_ok, int64_t S, int no_xnorm,
                                                                const int64_t* __restrict__ row_list = nullptr,
                                                                const int* __restrict__ row_count = nullptr) {
    constexpr int HALF = DP / 2;
    constexpr int LDE = DP + 4;
    extern __shared__ __align__(16) float smem[];
    float* s_e = smem;                       // [kTileCodes][LDE]
    float* s_ee = smem + kTileCodes * LDE;   // [kTileCodes]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    int64_t total = N;
    if constexpr (LIST) {
        const int64_t c = *row_count;
        total = c < N ? c : N;
    }
  for (int64_t vb = blockIdx.x; vb * kRowsPerBlock < total; vb += gridDim.x) {
    const int64_t slot = vb * kRowsPerBlock + wave * 32 + j;
    int64_t row = slot, row_c = slot < total ? slot : total - 1;
    if constexpr (LIST) {
        row_c = row_list[row_c];
        row_c = row_c < 0 ? 0 : (row_c < N ? row_c : N - 1);
        row = slot < total ? row_c : N;
    }

    // ---- B fragments: this lane's half of its input row, zero padded to DP
    float bx[HALF];
    if constexpr (LAYOUT == LBM_VQ_ROWS) {
        const float* p = x + row_c * ldx + h * HALF;
        if (vec_ok && DP == D) {
#pragma unroll
            for (int q = 0; q < HALF / 4; ++q) {
                const float4 v = reinterpret_cast<const float4*>(p)[q];
                bx[4 * q] = v.x; bx[4 * q + 1] = v.y; bx[4 * q + 2] = v.z; bx[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int s = 0; s < HALF; ++s) bx[s] = (h * HALF + s < D) ? p[s] : 0.f;
        }
    } else {
        const int64_t b = row_c / S, sp = row_c - b * S;   // ldx = pitch of the [b][d] rows of z, S positions per row
        const float* p = x + (b * D + h * HALF) * ldx + sp;
#pragma unroll
        for (int s = 0; s < HALF; ++s) bx[s] = (h * HALF + s < D) ? p[(int64_t)s * ldx] : 0.f;
    }
    float xx = 0.f;
#pragma unroll
    for (int s = 0; s < HALF; ++s) xx = fmaf(bx[s], bx[s], xx);
    xx += __shfl_xor(xx, 32, 64);
    if (no_xnorm) xx = 0.f;   // folded codebooks: the row constant is dropped (finer rounding grid for the scores)

    float best = INFINITY;
    int bidx = 0;
    const int d4 = D >> 2;  // float4 per codebook row
    for (int kbase = 0; kbase < K; kbase += kTileCodes) {
        __syncthreads();  // previous tile fully consumed
        // tile load: all loads of a group issued before the first LDS write (a rolled "load, wait, write"
        // loop costs one L2 round trip per float4); clamped address + mask instead of a predicated load
        constexpr int kF4 = kTileCodes * (DP / 4) / kThreads;   // float4 per thread per tile
        constexpr int kGroup = kF4 < 8 ? kF4 : 8;
        if (D == DP && kbase + kTileCodes <= K) {
            // full tile: 64 x DP contiguous floats, chunk i of the tile is chunk i of the LDS image (rows padded by 4):
            // one base address, immediate offsets, no masks (VALU slots are MFMA slots on this SIMD)
            const float4* tile = reinterpret_cast<const float4*>(codebook + (int64_t)kbase * D) + threadIdx.x;
#pragma unroll
            for (int g0 = 0; g0 < kF4; g0 += kGroup) {
                float4 stage[kGroup];
#pragma unroll
                for (int u = 0; u < kGroup; ++u) stage[u] = tile[(g0 + u) * kThreads];
#pragma unroll
                for (int u = 0; u < kGroup; ++u) {
                    const int i = threadIdx.x + (g0 + u) * kThreads;
                    const int code = i / (DP / 4), c4 = i - code * (DP / 4);
                    asm volatile("" : "+v"(stage[u].x), "+v"(stage[u].y), "+v"(stage[u].z), "+v"(stage[u].w));
                    *reinterpret_cast<float4*>(s_e + code * LDE + 4 * c4) = stage[u];
                }
            }
        } else
#pragma unroll
        for (int g0 = 0; g0 < kF4; g0 += kGroup) {
            float4 stage[kGroup];
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const int i = threadIdx.x + (g0 + u) * kThreads;
                const int code = i / (DP / 4), c4 = i - code * (DP / 4);
                const bool ok = kbase + code < K && c4 < d4;
                stage[u] = reinterpret_cast<const float4*>(codebook + (int64_t)(ok ? kbase + code : 0) * D)[ok ? c4 : 0];
            }
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const int i = threadIdx.x + (g0 + u) * kThreads;
                const int code = i / (DP / 4), c4 = i - code * (DP / 4);
                const bool ok = kbase + code < K && c4 < d4;
                asm volatile("" : "+v"(stage[u].x), "+v"(stage[u].y), "+v"(stage[u].z), "+v"(stage[u].w));
                *reinterpret_cast<float4*>(s_e + code * LDE + 4 * c4) = ok ? stage[u] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if (threadIdx.x < kTileCodes)
            s_ee[threadIdx.x] = (kbase + threadIdx.x < K) ? e_sqnorm[kbase + threadIdx.x] : INFINITY;
        __syncthreads();

        // Both 32-code halves of the tile at once, on two independent accumulators: a single dependent
        // chain of v_mfma_f32_32x32x2_f32 sustains only ~70 % of the issue rate (tools/probe/mfma_probe.hip,
        // mode 3: 107 TFLOP/s; two chains: ~140).  Each chain is still the same ascending fmaf order.  Codes
        // beyond K are zero rows with |e|^2 = +inf: their distance is +inf and never wins.
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.f;
            acc1[r] = 0.f;
        }
        const float* ap0 = s_e + j * LDE + h * HALF;
        const float* ap1 = ap0 + 32 * LDE;
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
            const float4 a0 = reinterpret_cast<const float4*>(ap0)[q];
            const float4 a1 = reinterpret_cast<const float4*>(ap1)[q];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bx[4 * q + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bx[4 * q + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bx[4 * q + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bx[4 * q + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bx[4 * q + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bx[4 * q + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bx[4 * q + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bx[4 * q + 3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float dist = (xx + s_ee[cl]) - 2.f * (c == 0 ? acc0[r] : acc1[r]);
                if (dist < best) {
                    best = dist;
                    bidx = kbase + cl;
                }
            }
        }
    }
    // merge the two lane halves of each input row: smaller distance, then smaller index
    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64);
    if (ob < best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
    }
    if (h == 0 && row < N) {
        idx_out[row] = (int64_t)bidx;
        if (mind != nullptr) mind[row] = best;
    }
    if constexpr (!LIST) break;   // one block of rows per workgroup
  }
}

// D = 256, K a multiple of 64, 16-byte aligned codebook: the same arithmetic (same ascending fmaf chains, same argmin
// rule) with the tile traffic off the critical path.  8 waves (256 input rows) share each 64-code tile; the tiles are
// double buffered in LDS and filled by LDS-DMA (global_load_lds_dwordx4: one wave-instruction = one 1 KiB code row
// dropped at its padded LDS row, no staging registers, no ds_write), tile t+1 requested before tile t's MFMA stream
// and awaited after it: one barrier per tile and no exposed L2 round trip.  |e|^2 of the tile rides the same way.
constexpr int kDmaThreads = 512;
constexpr int kDmaRows = 256;
constexpr int kDmaLDE = 256 + 4;
constexpr int kDmaBufFloats = kTileCodes * kDmaLDE + kTileCodes;

template <int LAYOUT>
__global__ __launch_bounds__(kDmaThreads) void vq_argmin_dma_kernel(const float* __restrict__ x, int64_t ldx,
                                                                    const float* __restrict__ codebook,
                                                                    const float* __restrict__ e_sqnorm,
                                                                    int64_t* __restrict__ idx_out, float* __restrict__ mind,
                                                                    int64_t N, int K, int vec_ok, int64_t S, int no_xnorm) {
    constexpr int D = 256, HALF = 128, LDE = kDmaLDE;
    extern __shared__ __align__(16) float smem[];   // [2][ 64 x LDE tile | 64 |e|^2 ]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int64_t row = (int64_t)blockIdx.x * kDmaRows + wave * 32 + j;
    const int64_t row_c = row < N ? row : N - 1;

    // tile -> LDS: wave w copies code rows 8w .. 8w+7 of the tile; wave 0 also the 64 |e|^2
    auto copy_tile = [&](int kbase, int buf) {
        float* dst = smem + buf * kDmaBufFloats;
        const float* src = codebook + (int64_t)(kbase + wave * 8) * D + lane * 4;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + r * D),
                                             (__attribute__((address_space(3))) void*)(dst + (wave * 8 + r) * LDE), 16, 0, 0);
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(e_sqnorm + kbase + lane),
                                             (__attribute__((address_space(3))) void*)(dst + kTileCodes * LDE), 4, 0, 0);
    };

    // ---- B fragments: this lane's half of its input row.  Only the first half of it is loaded before the first
    // barrier; the second half is requested at the top of the first tile and consumed by the second half of that
    // tile's MFMA stream (with one workgroup per CU nothing else would hide the row load).
    float bx[HALF];
    auto load_x = [&](int s0, int s1) {
        if constexpr (LAYOUT == LBM_VQ_ROWS) {
            const float* p = x + row_c * ldx + h * HALF;
            if (vec_ok) {
#pragma unroll
                for (int q = s0 / 4; q < s1 / 4; ++q) {
                    const float4 v = reinterpret_cast<const float4*>(p)[q];
                    bx[4 * q] = v.x; bx[4 * q + 1] = v.y; bx[4 * q + 2] = v.z; bx[4 * q + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int s = s0; s < s1; ++s) bx[s] = p[s];
            }
        } else {
            const int64_t b = row_c / S, sp = row_c - b * S;
            const float* p = x + (b * D + h * HALF) * ldx + sp;
            const int ld = (int)ldx;   // D * ldx < 2^31 is checked by the launcher
#pragma unroll
            for (int s = s0; s < s1; ++s) bx[s] = p[s * ld];
        }
    };
    load_x(0, HALF / 2);
    copy_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float xx = 0.f;

    float best = INFINITY;
    int bidx = 0;
    // one 64-code tile; the first call is peeled out of the loop below (hipcc drains vmcnt in a loop preheader when the
    // loop consumes registers loaded before it -- that would put the whole row load back in front of the first tile)
    auto do_tile = [&](const int kbase, const int buf) {
        if (kbase + kTileCodes < K) copy_tile(kbase + kTileCodes, buf ^ 1);   // its last readers passed the barrier below
        if (kbase == 0) {
            load_x(HALF / 2, HALF);
            __builtin_amdgcn_sched_barrier(0);   // all 64 requests go out here, not next to their uses
        }
        const float* s_e = smem + buf * kDmaBufFloats;
        const float* s_ee = s_e + kTileCodes * LDE;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = 0.f;
            acc1[r] = 0.f;
        }
        typedef float f4 __attribute__((ext_vector_type(4)));
        const unsigned char* ap0 = reinterpret_cast<const unsigned char*>(smem) + (buf * kDmaBufFloats + j * LDE + h * HALF) * 4;
        const unsigned char* ap1 = ap0 + 32 * LDE * 4;
#pragma unroll
        for (int q = 0; q < HALF / 4; ++q) {
            const f4 a0 = *reinterpret_cast<const f4*>(ap0 + 16 * q);
            const f4 a1 = *reinterpret_cast<const f4*>(ap1 + 16 * q);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bx[4 * q + 0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bx[4 * q + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bx[4 * q + 1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bx[4 * q + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bx[4 * q + 2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bx[4 * q + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bx[4 * q + 3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bx[4 * q + 3], acc1, 0, 0, 0);
        }
        if (kbase == 0 && !no_xnorm) {   // |x|^2, after the first tile's products so that it does not wait for the row up front
#pragma unroll
            for (int s = 0; s < HALF; ++s) xx = fmaf(bx[s], bx[s], xx);
            xx += __shfl_xor(xx, 32, 64);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = c * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float dist = (xx + s_ee[cl]) - 2.f * (c == 0 ? acc0[r] : acc1[r]);
                if (dist < best) {
                    best = dist;
                    bidx = kbase + cl;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the rows this wave copied for the next tile have landed
        __syncthreads();
    };
    do_tile(0, 0);
    for (int kbase = kTileCodes, buf = 1; kbase < K; kbase += kTileCodes, buf ^= 1) do_tile(kbase, buf);

    const float ob = __shfl_xor(best, 32, 64);
    const int oi = __shfl_xor(bidx, 32, 64);
    if (ob < best || (ob == best && oi < bidx)) {
        best = ob;
        bidx = oi;
    }
    if (h == 0 && row < N) {
        idx_out[row] = (int64_t)bidx;
        if (mind != nullptr) mind[row] = best;
    }
}

// usage histogram of int64 indices: LDS-privatised bins, one 64-bit atomic per non-empty bin per block
__global__ __launch_bounds__(256) void hist_kernel(const int64_t* __restrict__ idx, unsigned long long* __restrict__ hist,
                                                   int64_t N, int K) {
    extern __shared__ unsigned int s_h[];
    for (int i = threadIdx.x; i < K; i += 256) s_h[i] = 0u;
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        const int64_t k = idx[n];
        if (k >= 0 && k < K) atomicAdd(&s_h[(int)k], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < K; i += 256)
        if (s_h[i] != 0u) atomicAdd(&hist[i], (unsigned long long)s_h[i]);
}

__global__ __launch_bounds__(256) void rownorm2_kernel(const float* __restrict__ m, float* __restrict__ out, int64_t rows,
                                                       int D) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float v = m[row * (int64_t)D + d];
        s = fmaf(v, v, s);
    }
    s = lbm::wave_sum(s);
    if (lane == 0) out[row] = s;
}

// quantized = x + (E[idx] - x) (straight-through arithmetic), sq_err_sum += sum((E[idx]-x)^2) in float64
__global__ __launch_bounds__(256) void gather_loss_kernel(const float* __restrict__ x, const float* __restrict__ codebook,
                                                          const int64_t* __restrict__ idx, float* __restrict__ quantized,
                                                          double* __restrict__ sq_err_sum, int64_t N, int D, int K) {
    __shared__ double s_part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double err = 0.0;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < N; row += (int64_t)gridDim.x * 4) {
        int64_t k = idx[row];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        for (int d = lane; d < D; d += 64) {
            const float xv = x[row * (int64_t)D + d];
            const float diff = codebook[k * D + d] - xv;
            if (quantized != nullptr) quantized[row * (int64_t)D + d] = xv + diff;
            err = fma((double)diff, (double)diff, err);
        }
    }
    err = lbm::wave_sum(err);
    if (lane == 0) s_part[wave] = err;
    __syncthreads();
    if (threadIdx.x == 0 && sq_err_sum != nullptr) atomicAdd(sq_err_sum, (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
}

__global__ __launch_bounds__(256) void perplexity_kernel(const unsigned long long* __restrict__ hist, float* __restrict__ out,
                                                         int K) {
    __shared__ double s_part[4];
    __shared__ double s_total;
    double t = 0.0;
    for (int i = threadIdx.x; i < K; i += 256) t += (double)hist[i];
    t = lbm::wave_sum(t);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) s_total = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    __syncthreads();
    const float total = (float)s_total;
    double acc = 0.0;
    for (int i = threadIdx.x; i < K; i += 256) {
        const float p = (float)hist[i] / total;            // avg_probs (training/model.py:143)
        acc += (double)(p * logf(p + 1e-10f));               // fp32 term, float64 accumulation
    }
    acc = lbm::wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = expf(-(float)((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])));
}

template <int DP>
int launch_vq(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm, int64_t* idx_out,
              float* mind, int64_t N, int D, int K, hipStream_t st) {
    const size_t lds = ((size_t)kTileCodes * (DP + 4) + kTileCodes) * sizeof(float);
    const int64_t blocks = (N + kRowsPerBlock - 1) / kRowsPerBlock;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_vq_argmin_f32: grid too large");
    const int vec_ok = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0) ? 1 : 0;
    hipError_t e;
    if (layout == LBM_VQ_ROWS) {
        auto k = vq_argmin_kernel<DP, LBM_VQ_ROWS>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, ldx, codebook, e_sqnorm, idx_out, mind, N, D, K, vec_ok, S, no_xnorm,
                           (const int64_t*)nullptr, (const int*)nullptr);
    } else {
        auto k = vq_argmin_kernel<DP, LBM_VQ_CHANNEL_MAJOR>;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, ldx, codebook, e_sqnorm, idx_out, mind, N, D, K, vec_ok, S, no_xnorm,
                           (const int64_t*)nullptr, (const int*)nullptr);
    }
    return lbm::check_launch("lbm_vq_argmin_f32");
}

int launch_vq_dma(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                  int64_t* idx_out, float* mind, int64_t N, int K, hipStream_t st) {
    const size_t lds = (size_t)2 * kDmaBufFloats * sizeof(float);
    const int64_t blocks = (N + kDmaRows - 1) / kDmaRows;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_vq_argmin_f32: grid too large");
    const int vec_ok = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0) ? 1 : 0;
    auto k = layout == LBM_VQ_ROWS ? vq_argmin_dma_kernel<LBM_VQ_ROWS> : vq_argmin_dma_kernel<LBM_VQ_CHANNEL_MAJOR>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kDmaThreads), lds, st, x, ldx, codebook, e_sqnorm, idx_out, mind, N, K, vec_ok, S,
                       no_xnorm);
    return lbm::check_launch("lbm_vq_argmin_f32");
}

}  // namespace

namespace {
template <int DP>
int launch_refine(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                  int64_t* idx_out, const int64_t* row_list, const int* row_count, int64_t N, int D, int K, hipStream_t st) {
    const size_t lds = ((size_t)kTileCodes * (DP + 4) + kTileCodes) * sizeof(float);
    int64_t blocks = (N + kRowsPerBlock - 1) / kRowsPerBlock;
    if (blocks > 1024) blocks = 1024;
    const int vec_ok = (((uintptr_t)x & 15) == 0 && (ldx & 3) == 0) ? 1 : 0;
    auto k = layout == LBM_VQ_ROWS ? vq_argmin_kernel<DP, LBM_VQ_ROWS, true> : vq_argmin_kernel<DP, LBM_VQ_CHANNEL_MAJOR, true>;
    // (the opt-in to this much dynamic LDS is a per-function, per-device attribute: set once per device, not on every call)
    static std::atomic<bool> attr_done[2][64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    const int lay = layout == LBM_VQ_ROWS ? 0 : 1;
    if (dev < 0 || !attr_done[lay][dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin_screened_f32(refine): hipFuncSetAttribute: %s", hipGetErrorString(e));
        if (dev >= 0) attr_done[lay][dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, ldx, codebook, e_sqnorm, idx_out, (float*)nullptr, N, D, K, vec_ok,
                       S, no_xnorm, row_list, row_count);
    return lbm::check_launch("lbm_vq_argmin_screened_f32(refine)");
}
}  // namespace

// shared with vq_screen.hip: full fp32 search of the listed rows (list and count live in device memory), with the fragment
// width lbm_vq_argmin_f32 picks for this D -- the width fixes the order of the fmaf chain, and the point is the same bits
int lbm_vq_refine_launch(const float* x, int64_t ldx, int layout, int no_xnorm, int64_t S, const float* codebook, const float* e_sqnorm,
                         int64_t* idx_out, const int64_t* row_list, const int* row_count, int64_t N, int D, int K, hipStream_t st) {
    if (D <= 64) return launch_refine<64>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, row_list, row_count, N, D, K, st);
    if (D <= 128) return launch_refine<128>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, row_list, row_count, N, D, K, st);
    return launch_refine<256>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, row_list, row_count, N, D, K, st);
}

// shared with vq_bf16.hip
int lbm_vq_hist_launch(const int64_t* idx, unsigned long long* hist, int64_t N, int K, hipStream_t st) {
    int64_t blocks = (N + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = (size_t)K * sizeof(unsigned int);      // one private bin per code: 128 KiB at K = 32768
    if (lds > 48 * 1024) {                                    // beyond the default dynamic-LDS limit: opt in (160 KB per CU on gfx950)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_vq_argmin(hist): hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(hist_kernel, dim3((unsigned)blocks), dim3(256), lds, st, idx, hist, N, K);
    return lbm::check_launch("lbm_vq_argmin(hist)");
}

extern "C" int lbm_rownorm2_f32(const float* m, float* out, int64_t rows, int32_t D, lbm_stream_t stream) {
    LBM_REQUIRE(rows >= 0 && D >= 1, LBM_EINVAL, "lbm_rownorm2_f32: bad sizes");
    if (rows == 0) return LBM_OK;
    LBM_REQUIRE(m && out, LBM_EINVAL, "lbm_rownorm2_f32: NULL pointer");
    const int64_t blocks = (rows + 3) / 4;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_rownorm2_f32: grid too large");
    hipLaunchKernelGGL(rownorm2_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), m, out, rows, D);
    return lbm::check_launch("lbm_rownorm2_f32");
}

extern "C" int lbm_vq_argmin_f32(const float* x, int64_t ldx, int32_t layout, int64_t S, const float* codebook,
                                 const float* e_sqnorm, int64_t* idx_out, float* mind_or_null,
                                 unsigned long long* hist_or_null, int64_t N, int32_t D, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0, LBM_EINVAL, "lbm_vq_argmin_f32: negative N");
    LBM_REQUIRE(D >= 8 && D <= 256 && (D & 7) == 0, LBM_EUNSUPPORTED, "lbm_vq_argmin_f32: D=%d must be a multiple of 8 in [8,256]", D);
    LBM_REQUIRE(K >= 1, LBM_EINVAL, "lbm_vq_argmin_f32: K must be >= 1");
    LBM_REQUIRE(layout >= 0 && layout <= (LBM_VQ_CHANNEL_MAJOR | LBM_VQ_NO_XNORM), LBM_EINVAL, "lbm_vq_argmin_f32: unknown layout %d", layout);
    const int no_xnorm = (layout & LBM_VQ_NO_XNORM) ? 1 : 0;
    layout &= ~LBM_VQ_NO_XNORM;
    if (N == 0) return LBM_OK;
    LBM_REQUIRE(x && codebook && e_sqnorm && idx_out, LBM_EINVAL, "lbm_vq_argmin_f32: NULL pointer");
    LBM_REQUIRE(((uintptr_t)codebook & 15) == 0, LBM_EINVAL, "lbm_vq_argmin_f32: codebook must be 16-byte aligned");
    if (layout == LBM_VQ_ROWS)
        LBM_REQUIRE(ldx >= D, LBM_EINVAL, "lbm_vq_argmin_f32: ldx (%lld) < D (%d)", (long long)ldx, D);
    else
        LBM_REQUIRE(S >= 1 && ldx >= S && N % S == 0, LBM_EINVAL,
                    "lbm_vq_argmin_f32: channel-major needs 1 <= S <= ldx and N %% S == 0 (N=%lld, S=%lld, ldx=%lld)", (long long)N,
                    (long long)S, (long long)ldx);
    hipStream_t st = lbm::as_stream(stream);
    int rc;
    if (D <= 64) rc = launch_vq<64>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, mind_or_null, N, D, K, st);
    else if (D <= 128) rc = launch_vq<128>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, mind_or_null, N, D, K, st);
    else if (D == 256 && (K % kTileCodes) == 0 && ((uintptr_t)e_sqnorm & 3) == 0 &&
             (layout == LBM_VQ_ROWS || (int64_t)D * ldx < 0x7fffffffLL))
        rc = launch_vq_dma(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, mind_or_null, N, K, st);
    else rc = launch_vq<256>(x, ldx, layout, no_xnorm, S, codebook, e_sqnorm, idx_out, mind_or_null, N, D, K, st);
    if (rc != LBM_OK) return rc;
    if (hist_or_null != nullptr) {
        LBM_REQUIRE(K <= 32768, LBM_EUNSUPPORTED, "lbm_vq_argmin_f32: histogram supports K <= 32768");
        rc = lbm_vq_hist_launch(idx_out, hist_or_null, N, K, st);
    }
    return rc;
}

extern "C" int lbm_vq_gather_loss_f32(const float* x, const float* codebook, const int64_t* idx, float* quantized,
                                      double* sq_err_sum, int64_t N, int32_t D, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0 && D >= 1 && K >= 1, LBM_EINVAL, "lbm_vq_gather_loss_f32: bad sizes");
    if (N == 0) return LBM_OK;
    LBM_REQUIRE(x && codebook && idx, LBM_EINVAL, "lbm_vq_gather_loss_f32: NULL pointer");
    int64_t blocks = (N + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gather_loss_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), x, codebook, idx, quantized,
                       sq_err_sum, N, D, K);
    return lbm::check_launch("lbm_vq_gather_loss_f32");
}

namespace {

// idx[i] <- code_map[idx[i]]: tokens of a searched sub-codebook back to the caller's code numbers (16 bytes = two tokens per access;
// `head` = 1 when idx itself is only 8-byte aligned: its first token is then mapped alone and the pairs start behind it)
__global__ __launch_bounds__(256) void vq_remap_kernel(int64_t* __restrict__ idx, int64_t N, const int64_t* __restrict__ code_map, int Kc, int head) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t* body = idx + head;
    const int64_t nb = N - head, pairs = nb >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride) {
        longlong2 v = reinterpret_cast<longlong2*>(body)[i];
        v.x = (v.x >= 0 && v.x < Kc) ? code_map[v.x] : v.x;
        v.y = (v.y >= 0 && v.y < Kc) ? code_map[v.y] : v.y;
        reinterpret_cast<longlong2*>(body)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) {
            const int64_t v = idx[0];
            if (v >= 0 && v < Kc) idx[0] = code_map[v];
        }
        if (nb & 1) {
            const int64_t v = body[nb - 1];
            if (v >= 0 && v < Kc) body[nb - 1] = code_map[v];
        }
    }
}

// The remap above AND the usage histogram in one pass over the tokens: bins over the SEARCHED row (LDS-privatised, as hist_kernel),
// flushed as hist[code_map[k]] += bin[k].  The grid-stride walk is over 16-byte pairs; the odd head / tail tokens go with thread 0.
__global__ __launch_bounds__(256) void vq_remap_count_kernel(int64_t* __restrict__ idx, int64_t N, const int64_t* __restrict__ code_map, int Kc,
                                                             int head, unsigned long long* __restrict__ hist) {
    extern __shared__ unsigned int s_bins[];
    for (int i = threadIdx.x; i < Kc; i += 256) s_bins[i] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t* body = idx + head;
    const int64_t nb = N - head, pairs = nb >> 1;
    auto one = [&](int64_t v) -> int64_t {
        if (v < 0 || v >= Kc) return v;
        atomicAdd(&s_bins[(int)v], 1u);
        return code_map[v];
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += stride) {
        longlong2 v = reinterpret_cast<longlong2*>(body)[i];
        v.x = one(v.x);
        v.y = one(v.y);
        reinterpret_cast<longlong2*>(body)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (head) idx[0] = one(idx[0]);
        if (nb & 1) body[nb - 1] = one(body[nb - 1]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Kc; i += 256)
        if (s_bins[i] != 0u) atomicAdd(&hist[code_map[i]], (unsigned long long)s_bins[i]);
}

// hist[code_map[k]] += hist_c[k]; hist_c[k] <- 0.  code_map is injective on [0, Kc): no two k meet in one bin.
__global__ __launch_bounds__(256) void vq_remap_hist_kernel(unsigned long long* __restrict__ hist_c, unsigned long long* __restrict__ hist,
                                                            const int64_t* __restrict__ code_map, int Kc) {
    for (int k = blockIdx.x * 256 + threadIdx.x; k < Kc; k += gridDim.x * 256) {
        const unsigned long long c = hist_c[k];
        if (c != 0) {
            hist[code_map[k]] += c;
            hist_c[k] = 0;
        }
    }
}

}  // namespace

extern "C" int lbm_vq_remap_codes_i64(int64_t* idx, int64_t N, const int64_t* code_map, int32_t Kc, unsigned long long* hist_c_or_null,
                                      unsigned long long* hist_or_null, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0 && Kc >= 1, LBM_EINVAL, "lbm_vq_remap_codes_i64: bad sizes (N=%lld, Kc=%d)", (long long)N, Kc);
    LBM_REQUIRE(code_map, LBM_EINVAL, "lbm_vq_remap_codes_i64: NULL code_map");
    LBM_REQUIRE(hist_c_or_null == nullptr || hist_or_null != nullptr, LBM_EINVAL,
                "lbm_vq_remap_codes_i64: a searched-row histogram needs the full histogram to go to");
    hipStream_t st = lbm::as_stream(stream);
    if (hist_c_or_null == nullptr && hist_or_null != nullptr) {   // count the tokens of this call while they are remapped
        LBM_REQUIRE(Kc <= 12288, LBM_EUNSUPPORTED, "lbm_vq_remap_codes_i64: counting supports Kc <= 12288 (got %d): pass the searched-row histogram", Kc);
        if (N == 0) return LBM_OK;
        LBM_REQUIRE(idx, LBM_EINVAL, "lbm_vq_remap_codes_i64: NULL idx");
        LBM_REQUIRE(((uintptr_t)idx & 7) == 0, LBM_EINVAL, "lbm_vq_remap_codes_i64: idx must be 8-byte aligned");
        const int head = ((uintptr_t)idx & 15) ? 1 : 0;
        const int64_t want = ((N >> 1) + 255) / 256;
        const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
        hipLaunchKernelGGL(vq_remap_count_kernel, dim3(blocks), dim3(256), (size_t)Kc * sizeof(unsigned int), st, idx, N, code_map, Kc, head,
                           hist_or_null);
        return lbm::check_launch("lbm_vq_remap_codes_i64");
    }
    if (N > 0) {
        LBM_REQUIRE(idx, LBM_EINVAL, "lbm_vq_remap_codes_i64: NULL idx");
        LBM_REQUIRE(((uintptr_t)idx & 7) == 0, LBM_EINVAL, "lbm_vq_remap_codes_i64: idx must be 8-byte aligned");
        const int head = ((uintptr_t)idx & 15) ? 1 : 0;
        const int64_t want = ((N >> 1) + 255) / 256;
        const unsigned blocks = (unsigned)(want < 1 ? 1 : (want > 8192 ? 8192 : want));
        hipLaunchKernelGGL(vq_remap_kernel, dim3(blocks), dim3(256), 0, st, idx, N, code_map, Kc, head);
        const int rc = lbm::check_launch("lbm_vq_remap_codes_i64");
        if (rc != LBM_OK) return rc;
    }
    if (hist_or_null != nullptr) {
        hipLaunchKernelGGL(vq_remap_hist_kernel, dim3((unsigned)((Kc + 255) / 256)), dim3(256), 0, st, hist_c_or_null, hist_or_null, code_map, Kc);
        return lbm::check_launch("lbm_vq_remap_codes_i64");
    }
    return LBM_OK;
}

extern "C" int lbm_perplexity_u64(const unsigned long long* hist, float* out, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(K >= 1 && hist && out, LBM_EINVAL, "lbm_perplexity_u64: bad arguments");
    hipLaunchKernelGGL(perplexity_kernel, dim3(1), dim3(256), 0, lbm::as_stream(stream), hist, out, K);
    return lbm::check_launch("lbm_perplexity_u64");
}
