// EMA codebook training step of VectorQuantizerEMA (training/model.py:103-126), SURVEY.md 8(f) item 2.
//
// The reference forms one-hot encodings [N, K] and two matmuls with them:
//   counts = sum(encodings, 0)                      (:109-110)
//   dw     = encodings^T @ flat_input   [K, D]      (:122)
// Here counts is a histogram and dw a segmented sum of the input rows by code, deterministic (no float
// atomics): a stable counting sort of the row ids by code (block histograms -> per-code scan over blocks ->
// ballot-ranked scatter), then every code's run is cut into pieces of <= kPiece rows that are summed in row
// order by one workgroup each, and the pieces of a code are combined in piece order.  (counts, dw) are the
// quantities a data-parallel trainer all-reduces across ranks before lbm_vq_ema_apply_f32.
//   lbm_vq_ema_apply_f32: cluster_size = cluster_size*decay + (1-decay)*counts; n = sum(cluster_size);
//   cluster_size = (cluster_size + eps) / (n + K*eps) * n; ema_w = ema_w*decay + (1-decay)*dw;
//   weight = ema_w / cluster_size[:, None]   -- fp32, the reference's operation order.
#include "common.h"

namespace {

constexpr int kPiece = 4096;      // rows per reduction piece
constexpr int kMaxK = 8192;       // LDS counters per block

struct EmaPlan {
    int64_t rb;        // rows per sort block
    int64_t nblk;      // sort blocks
    int64_t pmax;      // upper bound of reduction pieces
    size_t off_bh, off_perm, off_start, off_pbase, off_partial, total;
};

EmaPlan make_plan(int64_t N, int D, int K) {
    EmaPlan p;
    p.rb = 2048;
    if (p.rb < 4LL * K) p.rb = 4LL * K;
    p.nblk = (N + p.rb - 1) / p.rb;
    if (p.nblk < 1) p.nblk = 1;
    p.pmax = N / kPiece + K;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 0;
    p.off_bh = o;      o = up(o + (size_t)p.nblk * K * sizeof(unsigned int));
    p.off_perm = o;    o = up(o + (size_t)(N > 0 ? N : 1) * sizeof(unsigned int));
    p.off_start = o;   o = up(o + (size_t)(K + 1) * sizeof(unsigned long long));
    p.off_pbase = o;   o = up(o + (size_t)(K + 1) * sizeof(unsigned int));
    p.off_partial = o; o = up(o + (size_t)p.pmax * D * sizeof(float));
    p.total = o;
    return p;
}

__global__ __launch_bounds__(256) void ema_blockhist_kernel(const int64_t* __restrict__ idx, unsigned int* __restrict__ bh,
                                                            int64_t N, int K, int64_t rb) {
    extern __shared__ unsigned int s_cnt[];
    for (int k = threadIdx.x; k < K; k += 256) s_cnt[k] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * rb, hi = lo + rb < N ? lo + rb : N;
    for (int64_t n = lo + threadIdx.x; n < hi; n += 256) {
        const int64_t k = idx[n];
        if (k >= 0 && k < K) atomicAdd(&s_cnt[(int)k], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < K; k += 256) bh[(int64_t)blockIdx.x * K + k] = s_cnt[k];
}

// per code: exclusive scan of its block counts (in place) and the total
__global__ __launch_bounds__(256) void ema_scan_kernel(unsigned int* __restrict__ bh, unsigned long long* __restrict__ counts,
                                                       int64_t nblk, int K) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    unsigned long long run = 0;
    for (int64_t b = 0; b < nblk; ++b) {
        const unsigned int v = bh[b * K + k];
        bh[b * K + k] = (unsigned int)run;
        run += v;
    }
    counts[k] = run;
}

// start[k] = first sorted position of code k; pbase[k] = first reduction piece of code k  (K is small: one thread)
__global__ void ema_starts_kernel(const unsigned long long* __restrict__ counts, unsigned long long* __restrict__ start,
                                  unsigned int* __restrict__ pbase, int K) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long s = 0;
    unsigned int p = 0;
    for (int k = 0; k < K; ++k) {
        start[k] = s;
        pbase[k] = p;
        s += counts[k];
        p += (unsigned int)((counts[k] + kPiece - 1) / kPiece);
    }
    start[K] = s;
    pbase[K] = p;
}

// one wavefront per sort block: rows in order, 64 at a time; equal codes are ranked with ballots, so the
// sorted order inside a code is ascending row id (stable)
__global__ __launch_bounds__(64) void ema_scatter_kernel(const int64_t* __restrict__ idx, const unsigned int* __restrict__ bh,
                                                         const unsigned long long* __restrict__ start,
                                                         unsigned int* __restrict__ perm, int64_t N, int K, int64_t rb,
                                                         int nbits) {
    extern __shared__ unsigned int s_run[];
    const int lane = threadIdx.x;
    for (int k = lane; k < K; k += 64) s_run[k] = (unsigned int)start[k] + bh[(int64_t)blockIdx.x * K + k];
    __builtin_amdgcn_wave_barrier();
    const int64_t lo = (int64_t)blockIdx.x * rb, hi = lo + rb < N ? lo + rb : N;
    const unsigned long long lt = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int64_t n0 = lo; n0 < hi; n0 += 64) {
        const int64_t n = n0 + lane;
        int64_t kk = n < hi ? idx[n] : -1;
        const bool valid = kk >= 0 && kk < K;
        const int code = valid ? (int)kk : 0;
        unsigned long long mask = __ballot(valid);
        for (int b = 0; b < nbits; ++b) {
            const unsigned long long bal = __ballot((code >> b) & 1);
            mask &= ((code >> b) & 1) ? bal : ~bal;
        }
        if (valid) {   // mask = lanes of this step that hold the same code (includes this lane)
            const int leader = __ffsll((long long)mask) - 1;
            unsigned int base = 0;
            if (lane == leader) {
                base = s_run[code];
                s_run[code] = base + (unsigned int)__popcll(mask);
            }
            base = __shfl(base, leader, 64);
            perm[base + (unsigned int)__popcll(mask & lt)] = (unsigned int)n;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// piece p of code k: rows perm[start[k] + j*kPiece ...) summed in order, thread d owns column d (+256, ...)
__global__ __launch_bounds__(256) void ema_piece_kernel(const float* __restrict__ x, int64_t ldx, const unsigned int* __restrict__ perm,
                                                        const unsigned long long* __restrict__ start,
                                                        const unsigned int* __restrict__ pbase, float* __restrict__ partial,
                                                        int D, int K) {
    __shared__ unsigned int s_rows[256];
    const unsigned int p = blockIdx.x;
    if (p >= pbase[K]) return;
    int a = 0, b = K;   // last k with pbase[k] <= p (empty codes share a pbase with their successor: take the last)
    while (b - a > 1) {
        const int m = (a + b) >> 1;
        if (pbase[m] <= p) a = m; else b = m;
    }
    const int k = a;
    const unsigned long long lo = start[k] + (unsigned long long)(p - pbase[k]) * kPiece;
    const unsigned long long end = start[k + 1];
    const unsigned long long hi = lo + kPiece < end ? lo + kPiece : end;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};   // columns threadIdx.x + 256*c  (D <= 1024)
    for (unsigned long long r0 = lo; r0 < hi; r0 += 256) {
        __syncthreads();
        if (r0 + threadIdx.x < hi) s_rows[threadIdx.x] = perm[r0 + threadIdx.x];
        __syncthreads();
        const int nr = (int)(hi - r0 < 256 ? hi - r0 : 256);
        int r = 0;
        for (; r + 4 <= nr; r += 4) {   // four rows in flight, added in row order
            const float* p0 = x + (int64_t)s_rows[r] * ldx;
            const float* p1 = x + (int64_t)s_rows[r + 1] * ldx;
            const float* p2 = x + (int64_t)s_rows[r + 2] * ldx;
            const float* p3 = x + (int64_t)s_rows[r + 3] * ldx;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int d = threadIdx.x + 256 * c;
                if (d < D) {
                    const float v0 = p0[d], v1 = p1[d], v2 = p2[d], v3 = p3[d];
                    acc[c] = (((acc[c] + v0) + v1) + v2) + v3;
                }
            }
        }
        for (; r < nr; ++r) {
            const float* p0 = x + (int64_t)s_rows[r] * ldx;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int d = threadIdx.x + 256 * c;
                if (d < D) acc[c] += p0[d];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int d = threadIdx.x + 256 * c;
        if (d < D) partial[(int64_t)p * D + d] = acc[c];
    }
}

__global__ __launch_bounds__(256) void ema_combine_kernel(const float* __restrict__ partial, const unsigned int* __restrict__ pbase,
                                                          float* __restrict__ dw, int D) {
    const int k = blockIdx.x;
    const unsigned int p0 = pbase[k], p1 = pbase[k + 1];
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (unsigned int p = p0; p < p1; ++p) acc += partial[(int64_t)p * D + d];
        dw[(int64_t)k * D + d] = acc;
    }
}

// one workgroup: EMA of the cluster sizes and Laplace smoothing (training/model.py:109-120)
__global__ __launch_bounds__(256) void ema_cluster_kernel(float* __restrict__ cs, const unsigned long long* __restrict__ counts, int K,
                                                          float decay, float omd, float eps) {
    __shared__ double s_part[4];
    __shared__ float s_n;
    double part = 0.0;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = cs[k] * decay + omd * (float)counts[k];
        cs[k] = v;
        part += (double)v;
    }
    part = lbm::wave_sum(part);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
    __syncthreads();
    if (threadIdx.x == 0) s_n = (float)((s_part[0] + s_part[1]) + (s_part[2] + s_part[3]));
    __syncthreads();
    const float n = s_n;
    const float den = n + (float)K * eps;
    for (int k = threadIdx.x; k < K; k += 256) cs[k] = (cs[k] + eps) / den * n;
}

__global__ __launch_bounds__(256) void ema_weight_kernel(float* __restrict__ ema_w, float* __restrict__ weight,
                                                         const float* __restrict__ dw, const float* __restrict__ cs, int64_t total,
                                                         int D, float decay, float omd) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float w = ema_w[i] * decay + omd * dw[i];
    ema_w[i] = w;
    weight[i] = w / cs[i / D];
}

}  // namespace

extern "C" size_t lbm_vq_ema_workspace_bytes(int64_t N, int32_t D, int32_t K) {
    if (N < 0 || D < 1 || K < 1 || K > kMaxK) return 0;
    return make_plan(N, D, K).total;
}

extern "C" int lbm_vq_ema_sums_f32(const float* x, int64_t ldx, const int64_t* idx, float* dw, unsigned long long* counts,
                                   void* workspace, int64_t N, int32_t D, int32_t K, lbm_stream_t stream) {
    LBM_REQUIRE(N >= 0 && D >= 1 && D <= 1024 && K >= 1, LBM_EINVAL, "lbm_vq_ema_sums_f32: bad sizes (N=%lld D=%d K=%d)", (long long)N, D, K);
    LBM_REQUIRE(K <= kMaxK, LBM_EUNSUPPORTED, "lbm_vq_ema_sums_f32: K=%d > %d", K, kMaxK);
    LBM_REQUIRE(N < 0xffffffffLL, LBM_EUNSUPPORTED, "lbm_vq_ema_sums_f32: N must be < 2^32");
    LBM_REQUIRE(dw && counts && workspace, LBM_EINVAL, "lbm_vq_ema_sums_f32: NULL pointer");
    LBM_REQUIRE(N == 0 || (x && idx), LBM_EINVAL, "lbm_vq_ema_sums_f32: NULL pointer");
    LBM_REQUIRE(N == 0 || ldx >= D, LBM_EINVAL, "lbm_vq_ema_sums_f32: ldx (%lld) < D (%d)", (long long)ldx, D);
    LBM_REQUIRE(((uintptr_t)workspace & 15) == 0, LBM_EINVAL, "lbm_vq_ema_sums_f32: workspace must be 16-byte aligned");
    const EmaPlan p = make_plan(N, D, K);
    char* ws = static_cast<char*>(workspace);
    unsigned int* bh = reinterpret_cast<unsigned int*>(ws + p.off_bh);
    unsigned int* perm = reinterpret_cast<unsigned int*>(ws + p.off_perm);
    unsigned long long* start = reinterpret_cast<unsigned long long*>(ws + p.off_start);
    unsigned int* pbase = reinterpret_cast<unsigned int*>(ws + p.off_pbase);
    float* partial = reinterpret_cast<float*>(ws + p.off_partial);
    hipStream_t st = lbm::as_stream(stream);
    int nbits = 0;
    while ((1 << nbits) < K) ++nbits;
    hipLaunchKernelGGL(ema_blockhist_kernel, dim3((unsigned)p.nblk), dim3(256), (size_t)K * sizeof(unsigned int), st, idx, bh, N, K, p.rb);
    hipLaunchKernelGGL(ema_scan_kernel, dim3((K + 255) / 256), dim3(256), 0, st, bh, counts, p.nblk, K);
    hipLaunchKernelGGL(ema_starts_kernel, dim3(1), dim3(64), 0, st, counts, start, pbase, K);
    hipLaunchKernelGGL(ema_scatter_kernel, dim3((unsigned)p.nblk), dim3(64), (size_t)K * sizeof(unsigned int), st, idx, bh, start, perm, N,
                       K, p.rb, nbits);
    hipLaunchKernelGGL(ema_piece_kernel, dim3((unsigned)p.pmax), dim3(256), 0, st, x, ldx, perm, start, pbase, partial, D, K);
    hipLaunchKernelGGL(ema_combine_kernel, dim3(K), dim3(256), 0, st, partial, pbase, dw, D);
    return lbm::check_launch("lbm_vq_ema_sums_f32");
}

extern "C" int lbm_vq_ema_apply_f32(float* ema_cluster_size, float* ema_w, float* weight, const float* dw,
                                    const unsigned long long* counts, int32_t K, int32_t D, float decay, float one_minus_decay,
                                    float eps, lbm_stream_t stream) {
    LBM_REQUIRE(K >= 1 && D >= 1, LBM_EINVAL, "lbm_vq_ema_apply_f32: bad sizes");
    LBM_REQUIRE(ema_cluster_size && ema_w && weight && dw && counts, LBM_EINVAL, "lbm_vq_ema_apply_f32: NULL pointer");
    hipStream_t st = lbm::as_stream(stream);
    hipLaunchKernelGGL(ema_cluster_kernel, dim3(1), dim3(256), 0, st, ema_cluster_size, counts, K, decay, one_minus_decay, eps);
    const int64_t total = (int64_t)K * D;
    hipLaunchKernelGGL(ema_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ema_w, weight, dw, ema_cluster_size,
                       total, D, decay, one_minus_decay);
    return lbm::check_launch("lbm_vq_ema_apply_f32");
}
