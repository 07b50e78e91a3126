// Welch PSD (a7) and band powers (a8).
//
// Tables (caller-owned device buffer, filled once by lbm_welch_tables_init in float64):
//   bytes 0..7   double  sum(w^2)
//   bytes 64..   float   window[nperseg4]            (periodic Hann / Hamming / boxcar)
//                float   cos[nfft4], sin[nfft4]      (twiddles of 2*pi*j/nfft)
//                float   basis[16][KD][4]            (folded DFT basis, only when nfft % 4 == 0 and nperseg == nfft)
//
// Fast path (welch_fold_kernel; nfft % 4 == 0, nperseg == nfft, <= 16 even and <= 16 odd bins, mean average):
//   two radix-2 decimation-in-frequency folds turn the N-point real DFT of the windowed segment into four
//   real (N/4+1)-point transforms -- even bins use ce[n] = a[n]+a[H-n], se[n] = b[n]-b[H-n], odd bins
//   co[n] = a[n]-a[H-n], so[n] = b[n]+b[H-n], with a[n] = w[n](d[n]+d[N-n]), b[n] = w[n](d[n]-d[N-n]),
//   H = N/2 -- and only the requested bins are evaluated: [16 segments x (N/4+1)] x [(N/4+1) x 16 bins]
//   products on v_mfma_f32_16x16x4_f32.  A wavefront stages a few whole rows in LDS (each sample read from
//   HBM once, coalesced), folds on the fly in registers straight into the MFMA A operand (no folded image
//   is ever stored), takes |X|^2 from the accumulators, and averages the row's segments through a small
//   LDS buffer.  4x fewer multiply-adds than the direct pruned DFT, no bank-conflicting twiddle gathers.
// Fallback (welch_kernel): one workgroup per row.  The row's segments are mean-removed, windowed and staged in
// LDS; each thread then evaluates the pruned DFT (only bins bin_lo..bin_hi, e.g. 27 of 251) of
// (segment, bin) pairs with an exact integer twiddle index (k*n mod nfft), and the segment average is
// taken in a fixed order.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kHeader = 64;  // bytes

__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }

struct TableView {
    const double* sum_w2;
    const float* window;
    const float* cosv;
    const float* sinv;
    const float* basis;
};

// folded-basis geometry for nfft % 4 == 0
__host__ __device__ inline int fold_qs(int nfft) { return (nfft / 4 + 1 + 3) / 4; }   // k-steps per lane quarter
__host__ __device__ inline int fold_kd(int nfft) { return 4 * fold_qs(nfft); }        // padded transform length
__host__ __device__ inline bool fold_ok_tables(int nfft, int nperseg) { return (nfft % 4) == 0 && nperseg == nfft && nfft >= 8; }
__host__ __device__ inline TableView view_tables(const void* tables, int nfft, int nperseg) {
    const char* p = reinterpret_cast<const char*>(tables);
    TableView v;
    v.sum_w2 = reinterpret_cast<const double*>(p);
    v.window = reinterpret_cast<const float*>(p + kHeader);
    v.cosv = v.window + pad4(nperseg);
    v.sinv = v.cosv + pad4(nfft);
    v.basis = v.sinv + pad4(nfft);
    return v;
}

__device__ inline double window_value(int window_id, int n, int N) {
    const double c = cospi(2.0 * (double)n / (double)N);
    if (window_id == LBM_WIN_HANN) return 0.5 - 0.5 * c;
    if (window_id == LBM_WIN_HAMMING) return 0.54 - 0.46 * c;
    return 1.0;
}

__global__ __launch_bounds__(kThreads) void welch_tables_kernel(void* tables, int nfft, int nperseg, int window_id,
                                                                int bin_lo, int bin_hi) {
    __shared__ double s_part[kThreads];
    char* p = reinterpret_cast<char*>(tables);
    float* window = reinterpret_cast<float*>(p + kHeader);
    float* cosv = window + pad4(nperseg);
    float* sinv = cosv + pad4(nfft);
    double acc = 0.0;
    for (int n = threadIdx.x; n < pad4(nperseg); n += kThreads) {
        const double w = (n < nperseg) ? window_value(window_id, n, nperseg) : 0.0;
        window[n] = (float)w;
        acc += w * w;  // float64 window, as scipy's scale uses
    }
    for (int j = threadIdx.x; j < pad4(nfft); j += kThreads) {
        const double a = 2.0 * (double)j / (double)nfft;
        cosv[j] = (j < nfft) ? (float)cospi(a) : 0.f;
        sinv[j] = (j < nfft) ? (float)sinpi(a) : 0.f;
    }
    if (fold_ok_tables(nfft, nperseg)) {
        // basis[j][n][v]: v = 0 cos (even bin), 1 sin (even bin), 2 cos (odd bin), 3 sin (odd bin); bin slot j.
        float* basis = sinv + pad4(nfft);
        const int KD = fold_kd(nfft), Q = nfft / 4;
        const int ke0 = bin_lo + (bin_lo & 1), ko0 = bin_lo + 1 - (bin_lo & 1);
        for (int i = threadIdx.x; i < 16 * KD; i += kThreads) {
            const int j = i / KD, n = i - j * KD;
            const int ke = ke0 + 2 * j, ko = ko0 + 2 * j;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n <= Q) {
                if (ke <= bin_hi) {
                    const double a = 2.0 * (double)(((long long)ke * n) % nfft) / (double)nfft;
                    v.x = (float)cospi(a);
                    v.y = (float)sinpi(a);
                }
                if (ko <= bin_hi) {
                    const double a = 2.0 * (double)(((long long)ko * n) % nfft) / (double)nfft;
                    v.z = (float)cospi(a);
                    v.w = (float)sinpi(a);
                }
            }
            reinterpret_cast<float4*>(basis)[i] = v;
        }
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < kThreads; ++i) s += s_part[i];
        *reinterpret_cast<double*>(p) = s;
        // header int at byte 8: 1 when the per-segment mean cannot reach the requested bins -- the DFT of a full-length periodic
        // Hann / Hamming window is zero outside bins 0 and +-1 (boxcar: outside bin 0), so sum_n w[n] m e^{-2 pi i k n / N} = 0
        // exactly for every requested k and the mean removal of scipy's detrend='constant' is a no-op there
        const bool mean_free = nperseg == nfft && ((window_id == LBM_WIN_BOXCAR && bin_lo >= 1) ||
                                                   ((window_id == LBM_WIN_HANN || window_id == LBM_WIN_HAMMING) && bin_lo >= 2));
        reinterpret_cast<int*>(p)[2] = mean_free ? 1 : 0;
    }
}

// ------------------------------------------------------------------------------------------ fast path
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr int kLdp = 33;  // row stride of the per-wave power buffer

struct FoldGeom {
    int N, H, Q, QS, KD, LDB;     // LDB: floats per bin row of the LDS basis image (KD*4 + pad, LDB/4 == 2 mod 16)
    int hop, nseg, RW, TILES;     // rows per group, 16-segment MFMA tiles per group
    int rows_floats;              // RW*T rounded up to 4
    int wave_floats;              // LDS floats per wave
};

__host__ __device__ inline FoldGeom fold_geom(int nfft, int T, int hop, int nseg) {
    FoldGeom g;
    g.N = nfft; g.H = nfft / 2; g.Q = nfft / 4; g.QS = fold_qs(nfft); g.KD = 4 * g.QS;
    int ld4 = g.KD;                       // in float4 units
    ld4 += ((2 - ld4) % 16 + 16) % 16;    // == 2 (mod 16): conflict-free ds_read_b128 for the (bin, kq) lane map
    g.LDB = 4 * ld4;
    g.hop = hop; g.nseg = nseg;
    g.RW = nseg >= 16 ? 1 : 16 / nseg;
    g.TILES = (g.RW * nseg + 15) / 16;
    g.rows_floats = (g.RW * T + 3) & ~3;
    g.wave_floats = g.rows_floats + g.TILES * 16 * kLdp + g.TILES * 16;
    return g;
}

__global__ __launch_bounds__(384) void welch_fold_kernel(const float* __restrict__ x, const float* __restrict__ row_scale,
                                                              float* __restrict__ psd, int64_t rows,
                                                              int T, double fs, FoldGeom g, const void* __restrict__ tables,
                                                              int bin_lo, int F, int64_t ngroups) {
    extern __shared__ __align__(16) float smem[];
    float* s_basis = smem;                       // [16][LDB]
    float* s_win = s_basis + 16 * g.LDB;         // [H+1] (padded to 4)
    const int win_floats = (g.H + 1 + 3) & ~3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* s_rows = s_win + win_floats + (size_t)wave * g.wave_floats;   // [RW][T]
    float* s_P = s_rows + g.rows_floats;                                 // [TILES*16][kLdp]
    float* s_mean = s_P + g.TILES * 16 * kLdp;                           // [TILES*16]

    const TableView tv = view_tables(tables, g.N, g.N);
    for (int i = threadIdx.x; i < 16 * g.KD; i += blockDim.x) {
        const int jj = i / g.KD, n = i - jj * g.KD;
        *reinterpret_cast<float4*>(s_basis + jj * g.LDB + 4 * n) = reinterpret_cast<const float4*>(tv.basis)[i];
    }
    for (int i = threadIdx.x; i <= g.H; i += blockDim.x) s_win[i] = tv.window[i];
    __syncthreads();   // the only workgroup barrier: waves are independent from here on

    const int j = lane & 15, kq = lane >> 4;
    const double scale = 1.0 / (fs * (*tv.sum_w2));
    const int waves_per_block = blockDim.x >> 6;
    const float inv_n = 1.f / (float)g.N;
    const float inv_F = 1.f / (float)F;
    const double scale2 = scale * 2.0;
    const int ke0 = bin_lo + (bin_lo & 1), ko0 = bin_lo + 1 - (bin_lo & 1);

    // Row staging.  With one wavefront per SIMD nothing hides an HBM round trip, so the NEXT group's rows are
    // loaded into registers (kPFQ quads per lane) while the current group is transformed, and only written to
    // LDS at the top of the next iteration.  Groups too large for the registers (or unaligned rows) use
    // batches of 8 loads in flight instead.
    constexpr int kPFQ = 24;
    const int64_t gstride = (int64_t)gridDim.x * waves_per_block;
    const bool use_pf = ((T & 3) == 0) && ((((uintptr_t)x) & 15) == 0) && (g.RW * T <= 4 * 64 * kPFQ);
    float4 pre[kPFQ];
    auto prefetch = [&](int64_t grp) {
        const int64_t row0 = grp * g.RW;
        const int nrows = (int)((rows - row0) < g.RW ? (rows - row0) : g.RW);
        const int n4 = (nrows * T) >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(x + row0 * (int64_t)T);
#pragma unroll
        for (int u = 0; u < kPFQ; ++u) {
            const int idx = u * 64 + lane;
            pre[u] = src4[idx < n4 ? idx : n4 - 1];
        }
    };
    const int64_t grp_first = (int64_t)blockIdx.x * waves_per_block + wave;
    if (use_pf && grp_first < ngroups) prefetch(grp_first);

    for (int64_t grp = grp_first; grp < ngroups; grp += gstride) {
        const int64_t row0 = grp * g.RW;
        const int nrows = (int)((rows - row0) < g.RW ? (rows - row0) : g.RW);
        const int nsegs = nrows * g.nseg;
        // ---- stage the group's rows (contiguous in x) into this wave's LDS
        if (use_pf) {
            const int n4 = (nrows * T) >> 2;
#pragma unroll
            for (int u = 0; u < kPFQ; ++u) {
                float4 v = pre[u];
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
                const int idx = u * 64 + lane;
                if (idx < n4) reinterpret_cast<float4*>(s_rows)[idx] = v;
            }
            const int64_t nxt = grp + gstride;
            prefetch(nxt < ngroups ? nxt : grp);   // flies under this group's transforms (the last one re-reads itself)
        } else {
            const float* src = x + row0 * (int64_t)T;
            const int nfl = nrows * T;
            if ((((uintptr_t)src) & 15) == 0) {
                // batches of 8 loads in flight per lane: a rolled "load, wait, write" loop pays one HBM round
                // trip per float4
                const int n4 = nfl >> 2;
                const float4* src4 = reinterpret_cast<const float4*>(src);
                for (int i0 = 0; i0 < n4; i0 += 64 * 8) {
                    float4 st[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = i0 + u * 64 + lane;
                        st[u] = src4[idx < n4 ? idx : n4 - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int idx = i0 + u * 64 + lane;
                        asm volatile("" : "+v"(st[u].x), "+v"(st[u].y), "+v"(st[u].z), "+v"(st[u].w));
                        if (idx < n4) reinterpret_cast<float4*>(s_rows)[idx] = st[u];
                    }
                }
                for (int i = (n4 << 2) + lane; i < nfl; i += 64) s_rows[i] = src[i];
            } else {
                for (int i = lane; i < nfl; i += 64) s_rows[i] = src[i];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- per-segment means: 4 lanes per segment, each sums every 4th group of 8 samples with 8 loads in
        // flight (a wave-wide reduction per segment costs 6 dependent cross-lane steps; one lane per segment
        // walking N samples pays an LDS round trip per step)
        for (int sg0 = 0; sg0 < nsegs; sg0 += 16) {
            const int sg = sg0 + (lane >> 2), part = lane & 3;
            const int sgc = sg < nsegs ? sg : nsegs - 1;
            const float* base = s_rows + (sgc / g.nseg) * T + (sgc % g.nseg) * g.hop;
            // the mean is kept as first sample + mean of the differences to it: rows with an offset far above the signal keep
            // their signal bits (s_mean holds the second term)
            const float b0 = base[0];
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int n = 8 * part;
            if (((T | g.hop) & 1) == 0) {   // 8-byte aligned segments: half the LDS instructions
                for (; n + 104 <= g.N; n += 128) {   // four groups per trip: 16 reads in flight, not 4
                    float2 v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float2*>(base + n + 32 * (u >> 2) + 2 * (u & 3));
#pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        acc[2 * (u & 3)] += v[u].x - b0;
                        acc[2 * (u & 3) + 1] += v[u].y - b0;
                    }
                }
                for (; n + 8 <= g.N; n += 32) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float2 v = *reinterpret_cast<const float2*>(base + n + 2 * u);
                        acc[2 * u] += v.x - b0;
                        acc[2 * u + 1] += v.y - b0;
                    }
                }
            } else {
                for (; n + 8 <= g.N; n += 32) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc[u] += base[n + u] - b0;
                }
            }
            if (n < g.N) {   // ragged tail group of this lane
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u] += (n + u < g.N) ? base[n + u] - b0 : 0.f;
            }
            float sum = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            if (part == 0 && sg < nsegs) s_mean[sg] = sum * inv_n;
        }
        __builtin_amdgcn_wave_barrier();
        // ---- folded pruned DFT, 16 segments per MFMA tile
        for (int tile = 0; tile < g.TILES; ++tile) {
            const int sg = tile * 16 + j;
            const int sgc = sg < nsegs ? sg : 0;
            const float* base = s_rows + (sgc / g.nseg) * T + (sgc % g.nseg) * g.hop;
            const float m = s_mean[sgc], b0 = base[0];   // every sample enters as (d - first sample) - m
            const float* bb = s_basis + j * g.LDB;
            f32x4 acc[4];   // re_e, im_e, re_o, im_o
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // operands of k-step t: the four folded samples, two window values, four basis values.  They are
            // loaded one step ahead (explicit software pipeline: the compiler does not unroll this runtime-length
            // loop, and with one wavefront per SIMD every LDS round trip would otherwise sit between the MFMAs)
            struct Ops { float d1, d2, d3, d4, w1, w2; float4 bs; int nc; };
            auto load_ops = [&](int t) {
                Ops o;
                const int n = 4 * t + kq;
                o.nc = n < g.Q ? n : g.Q;
                o.d1 = base[o.nc];
                o.d3 = base[g.H - o.nc];
                o.d2 = base[g.N - (o.nc ? o.nc : 1)];   // d[N-n]; d[N] does not exist -> 0 below
                o.d4 = base[g.H + o.nc];                // d[H+n]; at n == 0 it is d[H] again -> 0 below
                o.w1 = s_win[o.nc];
                o.w2 = s_win[g.H - o.nc];
                o.bs = *reinterpret_cast<const float4*>(bb + 4 * n);
                return o;
            };
            auto step_generic = [&](const Ops& cur) {
                const float d1 = (cur.d1 - b0) - m, d3 = (cur.d3 - b0) - m;
                float d2 = (cur.d2 - b0) - m, d4 = (cur.d4 - b0) - m;
                if (cur.nc == 0) { d2 = 0.f; d4 = 0.f; }
                const float a1 = cur.w1 * (d1 + d2), a2 = cur.w2 * (d3 + d4);
                const float b1 = cur.w1 * (d1 - d2), b2 = cur.w2 * (d3 - d4);
                const float f = (cur.nc == g.Q) ? 0.5f : 1.f;   // the middle term pairs with itself
                const float ce = f * (a1 + a2), co = a1 - a2, se = b1 - b2, so = f * (b1 + b2);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ce, cur.bs.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(se, cur.bs.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(co, cur.bs.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(so, cur.bs.w, acc[3], 0, 0, 0);
            };
            // Steps 1 .. QS-2 have 0 < n < Q on every lane: no clamps, no special terms, and the mean leaves the
            // difference terms -- a1 = w1 (d1 + d2 - 2m), b1 = w1 (d1 - d2).  They run two at a time from running
            // pointers (ascending ones at the pair's first step, descending ones at its second, so that both
            // steps are immediate offsets), each step's operands requested while the other step computes.
            struct Fast { float d1, d2, d3, d4, w1, w2; float4 bs; };
            const float m2 = 2.f * m;
            auto step_fast = [&](const Fast& o) {
                const float a1 = o.w1 * (((o.d1 - b0) + (o.d2 - b0)) - m2), a2 = o.w2 * (((o.d3 - b0) + (o.d4 - b0)) - m2);
                const float b1 = o.w1 * (o.d1 - o.d2), b2 = o.w2 * (o.d3 - o.d4);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1 + a2, o.bs.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1 - b2, o.bs.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1 - a2, o.bs.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1 + b2, o.bs.w, acc[3], 0, 0, 0);
            };
            step_generic(load_ops(0));
            int t = 1;
            const int t_end = g.QS - 1;
            if (t + 2 <= t_end) {
                const int n1 = 4 + kq;                      // n of step 1
                const float* p1 = base + n1;                // base[n]
                const float* p4 = base + g.H + n1;          // base[H + n]
                const float* pw1 = s_win + n1;              // win[n]
                const float* pb = bb + 4 * n1;              // basis[n]
                const float* q3 = base + g.H - n1 - 4;      // base[H - n] of the pair's second step
                const float* q2 = base + g.N - n1 - 4;      // base[N - n]
                const float* qw2 = s_win + g.H - n1 - 4;    // win[H - n]
                auto load_a = [&]() {
                    Fast o;
                    o.d1 = p1[0]; o.d4 = p4[0]; o.w1 = pw1[0]; o.d3 = q3[4]; o.d2 = q2[4]; o.w2 = qw2[4];
                    o.bs = *reinterpret_cast<const float4*>(pb);
                    return o;
                };
                auto load_b = [&]() {
                    Fast o;
                    o.d1 = p1[4]; o.d4 = p4[4]; o.w1 = pw1[4]; o.d3 = q3[0]; o.d2 = q2[0]; o.w2 = qw2[0];
                    o.bs = *reinterpret_cast<const float4*>(pb + 16);
                    return o;
                };
                Fast A = load_a();
                for (; t + 2 <= t_end; t += 2) {
                    const Fast B = load_b();
                    p1 += 8; p4 += 8; pw1 += 8; pb += 32; q3 -= 8; q2 -= 8; qw2 -= 8;
                    __builtin_amdgcn_sched_barrier(0);
                    step_fast(A);
                    A = load_a();   // the next pair's first step (after the last pair: an in-range read nobody uses)
                    __builtin_amdgcn_sched_barrier(0);
                    step_fast(B);
                }
            }
            for (; t < g.QS; ++t) step_generic(load_ops(t));
            // accumulator: column = bin slot j (lane & 15), rows = segments 4*kq + r
            const int ie = ke0 + 2 * j - bin_lo, io = ko0 + 2 * j - bin_lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* prow = s_P + (tile * 16 + 4 * kq + r) * kLdp;
                if (ie < F) prow[ie] = fmaf(acc[0][r], acc[0][r], acc[1][r] * acc[1][r]);
                if (io < F) prow[io] = fmaf(acc[2][r], acc[2][r], acc[3][r] * acc[3][r]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // ---- segment average (ascending segment order) and scaling; outputs of the group are contiguous
        const int nout = nrows * F;
        for (int o = lane; o < nout; o += 64) {
            const int r = (int)(((float)o + 0.5f) * inv_F), fb = o - r * F;   // o / F (exact: o < 16 F, F < 2^10)
            float acc = 0.f;
            for (int s = 0; s < g.nseg; ++s) acc += s_P[(r * g.nseg + s) * kLdp + fb];
            const int k = bin_lo + fb;
            double sc = (k == 0 || k == g.H) ? scale : scale2;
            if (row_scale != nullptr) {            // x is a per-row rescaled copy (z-scored epochs): PSD(s x) = s^2 PSD(x)
                const double rs = (double)row_scale[row0 + r];
                sc *= rs * rs;
            }
            psd[row0 * F + o] = (float)((double)(acc / (float)g.nseg) * sc);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------ team path (round 2)
// welch_team_kernel<TS, NTILES>: the arithmetic of welch_fold_kernel with the k-steps of a 16-segment tile SPLIT over the TS
// wavefronts of a workgroup (= one team) that share one staged row group.  The fold kernel is bound by instruction issue and
// LDS round trips at 1-2 waves per SIMD (phase stamps, round 2: stage 5.0 k / means 3.2 k / fold 10.6 k / average 2.1 k cycles
// per 3-row group and wave; the fold's 128 MFMAs are 4.1 k of that), and what limits its waves is LDS: 33-66 KB of basis per
// workgroup plus 18 KB (30 x 1500) to 40 KB (config 5's 10 000-sample rows) of staged rows PER WAVE.  Splitting the transform
// length over TS waves changes both terms: a wave now needs only 1/TS of the basis -- <= 16 k-steps x float4 = 64 VGPRs, so the
// basis lives in REGISTERS and leaves LDS (and the fold loses its one ds_read_b128 per step) -- and TS waves share one copy of
// the rows.  A workgroup is 25 KB (70 KB) of LDS: 6 teams of 2 = 12 waves per CU at 30 x 1500, 2 teams of 4 at config 5 (was 6
// and 2 waves).  Teams are separate workgroups, so their phases drift apart and one team's transfers overlap another's
// transforms; the barriers are team-local.  Rows reach LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no
// ds_write): the NEXT group is requested the moment the fold has finished with the row buffer and flies under the exchange /
// average / store of this one.  The partial transforms (4 accumulators per tile and wave) are summed through LDS by the tile's
// owner wave, in ascending wave order; the segment average and the store are spread over the team.  Where the per-segment
// mean cannot reach the requested bins (table header flag: full-length Hann / Hamming and bin_lo >= 2, boxcar and bin_lo >= 1)
// the means phase is gone: the fold subtracts the segment's FIRST SAMPLE instead (any constant is annihilated exactly there;
// this one keeps the sums well conditioned for rows with a large offset).  Where it is needed, the mean is carried as first
// sample + mean of the differences to it.
constexpr int kTeamSteps = 16;  // k-steps per wave held in registers (QS / TS rounded up must not exceed it)

__host__ __device__ inline int team_rows_pad(int rows_floats) { return (rows_floats + 255) & ~255; }   // whole 1 KiB DMA units

// KEEP (lbm_welch_keep_f32): the peak-to-peak reject test of the epoch post-processing (keep[row / C] = 0 where max - min > thresh) rides
// along -- the rows are in LDS anyway -- so the features-only branch needs no separate pass over the epochs.
template <int TS, int NTILES, bool KEEP = false>
__global__ __launch_bounds__(64 * TS) __attribute__((amdgpu_waves_per_eu((TS == 2 && NTILES == 1) ? 3 : 2, (TS == 2 && NTILES == 1) ? 3 : 2))) void welch_team_kernel(
    const float* __restrict__ x, const float* __restrict__ row_scale, float* __restrict__ psd, int64_t rows, int T, double fs, FoldGeom g,
    const void* __restrict__ tables, int bin_lo, int F, int64_t ngroups, uint8_t* __restrict__ keep = nullptr, int C = 1, float ptp_thresh = 0.f) {
    extern __shared__ __align__(16) float smem[];
    float* s_rows = smem;                                            // [RW][T], padded to whole DMA units
    float4* s_part = reinterpret_cast<float4*>(s_rows + team_rows_pad(g.rows_floats));   // [NTILES][TS-1][4][64] float4
    float* s_P = reinterpret_cast<float*>(s_part + NTILES * (TS - 1) * 4 * 64);          // [NTILES*16][kLdp]
    float* s_mean = s_P + NTILES * 16 * kLdp;                        // [NTILES*16]
    float* s_win = s_mean + NTILES * 16;                             // [H+1]
    float* s_mm = s_win + ((g.H + 1 + 3) & ~3);                       // KEEP: [RW][TS][2] min / max partials of the group's rows
    const int lane = threadIdx.x & 63, tw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // tw in an SGPR: its branches are scalar
    const int j = lane & 15, kq = lane >> 4;

    // rows of a group -> LDS: unit u = 64 lanes x 16 bytes; the calling wave moves units first, first + stride, ...  Lanes past
    // the end of the group re-read its last quad into the padding of the row buffer.
    auto request_rows = [&](int64_t grp, int first, int stride) {
        const int64_t row0 = grp * g.RW;
        const int nrows = (int)((rows - row0) < g.RW ? (rows - row0) : g.RW);
        const int n4 = (nrows * T) >> 2;
        const float4* src4 = reinterpret_cast<const float4*>(x + row0 * (int64_t)T);
        const int units = (n4 + 63) >> 6;
        for (int u = first; u < units; u += stride) {
            const int idx = u * 64 + lane;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src4 + (idx < n4 ? idx : n4 - 1)),
                                             (__attribute__((address_space(3))) void*)(s_rows + u * 256), 16, 0, 0);
        }
    };
    if ((int64_t)blockIdx.x < ngroups) request_rows(blockIdx.x, tw, TS);

    const TableView tv = view_tables(tables, g.N, g.N);
    const int mean_free = reinterpret_cast<const int*>(tables)[2];
    for (int i = threadIdx.x; i <= g.H; i += 64 * TS) s_win[i] = tv.window[i];
    // this wave's k-steps [t_lo, t_hi) and its slice of the folded basis (bin slot j, sample n = 4 t + kq), in registers
    const int t_lo = (tw * g.QS) / TS, t_hi = ((tw + 1) * g.QS) / TS;
    float4 breg[kTeamSteps];
#pragma unroll
    for (int st = 0; st < kTeamSteps; ++st) {
        const int t = t_lo + st < g.QS ? t_lo + st : g.QS - 1;
        breg[st] = reinterpret_cast<const float4*>(tv.basis)[j * g.KD + 4 * t + kq];
    }

    // window slices of the fast steps, in registers: {w[n], w[H - n]}; ZERO for the steps of the straight line the wave does
    // not own (t >= t_hi) and for the two end steps (general form below): those then add exact zeros
    f32x2 wv[kTeamSteps];
#pragma unroll
    for (int st = 0; st < kTeamSteps; ++st) {
        const int t = t_lo + st, n = 4 * t + kq;
        const bool own = t < t_hi && t != 0 && 4 * t + 3 < g.Q;
        const int nc = own ? n : 1;
        const float w1 = tv.window[nc], w2 = tv.window[g.H - nc];
        wv[st] = own ? (f32x2){w1, w2} : (f32x2){0.f, 0.f};
    }
    const float4 b_last = reinterpret_cast<const float4*>(tv.basis)[j * g.KD + 4 * (g.QS - 1) + kq];

    const double scale = 1.0 / (fs * (*tv.sum_w2));
    const double scale2 = scale * 2.0;
    const float inv_n = 1.f / (float)g.N;
    const float inv_F = 1.f / (float)F;
    const int ke0 = bin_lo + (bin_lo & 1), ko0 = bin_lo + 1 - (bin_lo & 1);

    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t row0 = grp * g.RW;
        const int nrows = (int)((rows - row0) < g.RW ? (rows - row0) : g.RW);
        const int nsegs = nrows * g.nseg;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // B: this group's rows have landed (first group: the window too)
        if constexpr (KEEP) {   // min / max of every row of the group: 16-byte LDS reads, lanes strided over the row, one shuffle tree per row
            const int q4 = T >> 2;
            for (int r = 0; r < nrows; ++r) {
                const float4* rp = reinterpret_cast<const float4*>(s_rows + r * T);
                float mn = INFINITY, mx = -INFINITY;
                for (int q = threadIdx.x; q < q4; q += 64 * TS) {
                    const float4 v = rp[q];
                    mn = fminf(fminf(mn, fminf(v.x, v.y)), fminf(v.z, v.w));
                    mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
                }
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) {
                    mn = fminf(mn, __shfl_xor(mn, d, 64));
                    mx = fmaxf(mx, __shfl_xor(mx, d, 64));
                }
                if (lane == 0) {
                    s_mm[(r * TS + tw) * 2] = mn;
                    s_mm[(r * TS + tw) * 2 + 1] = mx;
                }
            }
        }
        // ---- per-segment means (scipy detrend='constant') -- only where the mean can reach the requested bins
        if (!mean_free) {   // uniform
            for (int sg0 = 16 * tw; sg0 < NTILES * 16; sg0 += 16 * TS) {
                const int sg = sg0 + (lane >> 2), part = lane & 3;
                const int sgc = sg < nsegs ? sg : nsegs - 1;
                const float* base = s_rows + (sgc / g.nseg) * T + (sgc % g.nseg) * g.hop;
                const float b0 = base[0];      // the mean is kept as first sample + mean of the differences: rows with an offset
                float sum = 0.f;               // far above the signal keep their signal bits
                for (int n = part; n < g.N; n += 4) sum += base[n] - b0;
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                if (part == 0 && sg < nsegs) s_mean[sg] = sum * inv_n;
            }
            __syncthreads();
        }
        // ---- folded pruned DFT: this wave's k-steps of every tile.  Steps with 0 < n < Q on every lane ("fast": all but
        // t = 0 and t = QS - 1) run as ONE straight line of kTeamSteps steps from immediate LDS offsets -- no branches, no
        // clamps, window and basis from registers, packed f32 arithmetic: the VALU port, not the MFMA pipe, was what the
        // fold kernel saturated.  Steps of that line the wave does not own carry zero weights.  The two end steps take the
        // general form under a uniform branch.
        f32x4 accs[NTILES][4];   // [tile][re_e, im_e, re_o, im_o]
#pragma unroll
        for (int tile = 0; tile < NTILES; ++tile) {
            const int sg = tile * 16 + j;
            const int sgc = sg < nsegs ? sg : 0;
            const float* base = s_rows + (sgc / g.nseg) * T + (sgc % g.nseg) * g.hop;
            // every sample enters as (d - first sample) - r, r = mean of those differences (0 where the mean cannot reach the
            // requested bins): differences of neighbouring samples are (nearly) exact, whatever the row's offset
            const float b0 = base[0];
            const float rm = mean_free ? 0.f : s_mean[sgc];
            const float m2 = 2.f * rm;
            f32x4 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            // opaque copy of the lane's k offset: keeps the operand addresses from being hoisted out of the group loop as
            // loop invariants (they would occupy registers there for nothing)
            int kqv = kq;
            asm volatile("" : "+v"(kqv));
            auto generic = [&](int t, const float4 bs) {
                const int n = 4 * t + kqv;
                const int nc = n < g.Q ? n : g.Q;
                const float d1 = (base[nc] - b0) - rm, d3 = (base[g.H - nc] - b0) - rm;
                float d2 = (base[g.N - (nc ? nc : 1)] - b0) - rm;   // d[N - n]; d[N] does not exist
                float d4 = (base[g.H + nc] - b0) - rm;              // d[H + n]; at n == 0 it is d[H] again
                if (nc == 0) { d2 = 0.f; d4 = 0.f; }
                const float w1 = n <= g.Q ? s_win[nc] : 0.f, w2 = n <= g.Q ? s_win[g.H - nc] : 0.f;   // k-slots past the middle term
                const float a1 = w1 * (d1 + d2), a2 = w2 * (d3 + d4);
                const float b1 = w1 * (d1 - d2), b2 = w2 * (d3 - d4);
                const float f = (nc == g.Q) ? 0.5f : 1.f;   // the middle term pairs with itself
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(f * (a1 + a2), bs.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b1 - b2, bs.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1 - a2, bs.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(f * (b1 + b2), bs.w, acc[3], 0, 0, 0);
            };
            const int n0 = 4 * t_lo + kqv;                  // this lane's sample at st = 0
            const float* p1 = base + n0;                    // d[n]      at p1[4 st]
            const float* p4 = base + g.H + n0;              // d[H + n]  at p4[4 st]
            const float* p3 = base + g.H - n0;              // d[H - n]  at p3[-4 st]
            const float* p2 = base + g.N - n0;              // d[N - n]  at p2[-4 st]
            struct Ld { f32x2 da, db; };                    // {d1, d3}, {d2, d4}
            auto load = [&](int st) {
                Ld o;
                o.da.x = p1[4 * st]; o.db.y = p4[4 * st]; o.da.y = p3[-4 * st]; o.db.x = p2[-4 * st];
                return o;
            };
            const f32x2 m2v = {m2, m2}, b0v = {b0, b0};
            auto fast = [&](const Ld& o, const f32x2 w, const float4 bs) {
                const f32x2 a = w * (((o.da - b0v) + (o.db - b0v)) - m2v);  // {a1, a2}
                const f32x2 b = w * (o.da - o.db);          // {b1, b2}
                // the four A operands first, each in its own register (formed one by one in front of their MFMAs they all went through
                // ONE register, and every v_add then waited for the previous MFMA to have read it: round 3, conv_wino_ws.hip)
                float o0 = a.x + a.y, o1 = b.x - b.y, o2 = a.x - a.y, o3 = b.x + b.y;
                asm volatile("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o0, bs.x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o1, bs.y, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(o2, bs.z, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(o3, bs.w, acc[3], 0, 0, 0);
            };
            // st = 0: t = 0 on wave 0 (there d[N - n] of lane n = 0 would be another segment's sample: never read it)
            if (t_lo == 0) generic(0, breg[0]);             // uniform
            else fast(load(0), wv[0], breg[0]);
            Ld cur = load(1);
#pragma unroll
            for (int st = 1; st < kTeamSteps; ++st) {
                Ld nxt = cur;
                if (st + 1 < kTeamSteps) nxt = load(st + 1);
                __builtin_amdgcn_sched_barrier(0);
                fast(cur, wv[st], breg[st]);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
            if (t_hi == g.QS) generic(g.QS - 1, b_last);    // uniform: the last wave owns the middle term
#pragma unroll
            for (int a = 0; a < 4; ++a) accs[tile][a] = acc[a];
        }
        __syncthreads();   // C: every wave of the team is done reading the rows (and, from the previous group, P)
        if constexpr (KEEP) {
            if ((int)threadIdx.x < nrows) {   // one thread per row combines the waves' partials (rewritten only after the next barrier B)
                float mn = INFINITY, mx = -INFINITY;
#pragma unroll
                for (int w = 0; w < TS; ++w) {
                    mn = fminf(mn, s_mm[(threadIdx.x * TS + w) * 2]);
                    mx = fmaxf(mx, s_mm[(threadIdx.x * TS + w) * 2 + 1]);
                }
                if (mx - mn > ptp_thresh) keep[(row0 + threadIdx.x) / C] = 0;
            }
        }
        // ---- the NEXT group's rows are requested as soon as the row buffer is free; they land under the rest of this
        // iteration.  An LDS-DMA piece costs its wave ~200 cycles of issue here (stamps), so the pieces are spread: over both
        // waves of a team of two; in a team of four over the waves that own no tile, while the owners sum (after barrier D).
        constexpr bool kSpareWaves = TS > 2 && NTILES < TS;
        if (!kSpareWaves) {
            const int64_t nxt = grp + gridDim.x;
            if (nxt < ngroups) request_rows(nxt, tw, TS);   // uniform
        }
        // ---- partial transforms of the non-owner waves -> LDS.  Tile `tile` is owned by wave tile % TS.
#pragma unroll
        for (int tile = 0; tile < NTILES; ++tile) {
            const int owner = tile % TS;
            if (tw != owner) {   // uniform
                const int slot = tw < owner ? tw : tw - 1;
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    s_part[((tile * (TS - 1) + slot) * 4 + a) * 64 + lane] =
                        make_float4(accs[tile][a][0], accs[tile][a][1], accs[tile][a][2], accs[tile][a][3]);
            }
        }
        __syncthreads();   // D
        if (kSpareWaves && tw >= NTILES) {   // uniform
            const int64_t nxt = grp + gridDim.x;
            if (nxt < ngroups) request_rows(nxt, tw - NTILES, TS - NTILES);
        }
        // ---- the owner sums the partials in ascending wave order, |X|^2 -> P.  Accumulator layout: column = bin slot j
        // (lane & 15), rows = segments 4 kq + r
#pragma unroll
        for (int tile = 0; tile < NTILES; ++tile) {
            const int owner = tile % TS;
            if (tw == owner) {   // uniform
                float4 sum[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) sum[a] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int w = 0; w < TS; ++w) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float4 v;
                        if (w == owner) {
                            v = make_float4(accs[tile][a][0], accs[tile][a][1], accs[tile][a][2], accs[tile][a][3]);
                        } else {
                            v = s_part[((tile * (TS - 1) + (w < owner ? w : w - 1)) * 4 + a) * 64 + lane];
                        }
                        sum[a].x += v.x; sum[a].y += v.y; sum[a].z += v.z; sum[a].w += v.w;
                    }
                }
#pragma unroll
                for (int odd = 0; odd < 2; ++odd) {
                    const float4 re = sum[2 * odd], im = sum[2 * odd + 1];
                    const int ib = (odd ? ko0 : ke0) + 2 * j - bin_lo;
                    if (ib < F) {
                        float* prow = s_P + (tile * 16 + 4 * kq) * kLdp + ib;
                        prow[0] = fmaf(re.x, re.x, im.x * im.x);
                        prow[kLdp] = fmaf(re.y, re.y, im.y * im.y);
                        prow[2 * kLdp] = fmaf(re.z, re.z, im.z * im.z);
                        prow[3 * kLdp] = fmaf(re.w, re.w, im.w * im.w);
                    }
                }
            }
        }
        __syncthreads();   // E: P complete
        // ---- segment average (ascending segment order), scaling, store: spread over the team
        const int nout = nrows * F;
        for (int o = threadIdx.x; o < nout; o += 64 * TS) {
            const int r = (int)(((float)o + 0.5f) * inv_F), fb = o - r * F;   // o / F (exact: o < 16 F, F < 2^10)
            float acc = 0.f;
            for (int sgi = 0; sgi < g.nseg; ++sgi) acc += s_P[(r * g.nseg + sgi) * kLdp + fb];
            const int k = bin_lo + fb;
            double sc = (k == 0 || k == g.H) ? scale : scale2;
            if (row_scale != nullptr) {
                const double rs = (double)row_scale[row0 + r];
                sc *= rs * rs;
            }
            psd[row0 * F + o] = (float)((double)(acc / (float)g.nseg) * sc);
        }
        // the next group writes the partials after its barrier C and P after its barrier D: every wave has left this average
        // by then
    }
}

__global__ __launch_bounds__(kThreads) void welch_kernel(const float* __restrict__ x, const float* __restrict__ row_scale,
                                                         float* __restrict__ psd,
                                                         float* __restrict__ seg_out, int T, double fs, int nfft,
                                                         int nperseg, int hop, int nseg, const void* __restrict__ tables,
                                                         int bin_lo, int F, int average_mode) {
    extern __shared__ __align__(16) float smem[];
    float* s_cos = smem;                          // [nfft]
    float* s_sin = s_cos + pad4(nfft);            // [nfft]
    float* s_xw = s_sin + pad4(nfft);             // [nseg][nperseg]
    float* s_p = s_xw + (size_t)nseg * nperseg;   // [nseg][F]
    const TableView tv = view_tables(tables, nfft, nperseg);
    const int64_t row = blockIdx.x;
    const float* xr = x + row * (int64_t)T;

    for (int j = threadIdx.x; j < nfft; j += kThreads) {
        s_cos[j] = tv.cosv[j];
        s_sin[j] = tv.sinv[j];
    }
    // per-segment mean removal + window: one wavefront per segment
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s = wave; s < nseg; s += kThreads / 64) {
        const float* xs = xr + (int64_t)s * hop;
        float sum = 0.f;
        for (int n = lane; n < nperseg; n += 64) sum += xs[n];
        const float mean = lbm::wave_sum(sum) / (float)nperseg;
        for (int n = lane; n < nperseg; n += 64) s_xw[(size_t)s * nperseg + n] = (xs[n] - mean) * tv.window[n];
    }
    __syncthreads();

    double scale = 1.0 / (fs * (*tv.sum_w2));
    if (row_scale != nullptr) scale *= (double)row_scale[row] * (double)row_scale[row];
    const int nout = nseg * F;
    for (int o = threadIdx.x; o < nout; o += kThreads) {
        const int s = o / F, k = bin_lo + (o % F);
        const float* xs = s_xw + (size_t)s * nperseg;
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < nperseg; ++n) {
            const float v = xs[n];
            re = fmaf(v, s_cos[idx], re);
            im = fmaf(v, s_sin[idx], im);
            idx += k;
            if (idx >= nfft) idx -= nfft;
        }
        const bool one_sided = (k == 0) || ((nfft & 1) == 0 && k == nfft / 2);
        const double p = ((double)re * re + (double)im * im) * scale * (one_sided ? 1.0 : 2.0);
        s_p[o] = (float)p;
    }
    __syncthreads();

    for (int b = threadIdx.x; b < F; b += kThreads) {
        if (seg_out != nullptr)
            for (int s = 0; s < nseg; ++s) seg_out[(row * F + b) * (int64_t)nseg + s] = s_p[s * F + b];
        if (average_mode == LBM_AVG_MEAN) {
            float acc = 0.f;
            for (int s = 0; s < nseg; ++s) acc += s_p[s * F + b];
            psd[row * F + b] = acc / (float)nseg;
        } else if (average_mode == LBM_AVG_MEDIAN) {
            // rank selection without reordering; ties resolved by position
            float lo = 0.f, hi = 0.f;
            const int r_lo = (nseg - 1) / 2, r_hi = nseg / 2;
            for (int s = 0; s < nseg; ++s) {
                const float v = s_p[s * F + b];
                int rank = 0;
                for (int u = 0; u < nseg; ++u) {
                    const float w = s_p[u * F + b];
                    rank += (w < v) || (w == v && u < s);
                }
                if (rank == r_lo) lo = v;
                if (rank == r_hi) hi = v;
            }
            double bias = 1.0;  // scipy.signal._spectral_py._median_bias
            for (int i = 2; i <= nseg; ++i) bias += ((i & 1) ? 1.0 : -1.0) / (double)i;
            psd[row * F + b] = (float)((0.5 * ((double)lo + (double)hi)) / bias);
        }
    }
}

// one wavefront per epoch (or per (epoch, channel) row when per_channel)
__global__ __launch_bounds__(256) void bandpower_kernel(const float* __restrict__ psd, const float* __restrict__ freqs,
                                                        float* __restrict__ out, int64_t groups, int n_elem, int F,
                                                        int already_db) {
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int lane = threadIdx.x & 63;
    const float* p = psd + g * (int64_t)n_elem;
    double st = 0.0, sa = 0.0, sb = 0.0;  // float64 accumulation of the fp32 dB values
    int nt = 0, na = 0, nb = 0;
    for (int i = lane; i < n_elem; i += 64) {
        const float f = freqs[i % F];
        const double v = already_db ? (double)p[i] : (double)(10.f * log10f(p[i]));
        if (f >= 4.f && f < 8.f) { st += v; ++nt; }
        else if (f >= 8.f && f < 13.f) { sa += v; ++na; }
        else if (f >= 13.f && f < 30.f) { sb += v; ++nb; }
    }
    st = lbm::wave_sum(st); sa = lbm::wave_sum(sa); sb = lbm::wave_sum(sb);
    const double ct = lbm::wave_sum((double)nt), ca = lbm::wave_sum((double)na), cb = lbm::wave_sum((double)nb);
    if (lane == 0) {
        const double theta = st / ct, alpha = sa / ca, beta = sb / cb;
        float* o = out + g * 6;
        o[0] = (float)theta; o[1] = (float)alpha; o[2] = (float)beta;
        o[3] = (float)(beta / theta); o[4] = (float)(alpha / beta); o[5] = (float)(theta / beta);
    }
}

size_t welch_lds_bytes(int nfft, int nperseg, int nseg, int F) {
    return ((size_t)2 * pad4(nfft) + (size_t)nseg * nperseg + (size_t)nseg * F) * sizeof(float);
}

}  // namespace

extern "C" size_t lbm_welch_tables_bytes(int32_t nfft, int32_t nperseg, int32_t bin_lo, int32_t bin_hi) {
    (void)bin_lo; (void)bin_hi;
    if (nfft < 1 || nperseg < 1) return 0;
    size_t floats = (size_t)pad4(nperseg) + 2 * (size_t)pad4(nfft);
    if (fold_ok_tables(nfft, nperseg)) floats += (size_t)16 * fold_kd(nfft) * 4;
    return (size_t)kHeader + floats * sizeof(float);
}

extern "C" int lbm_welch_tables_init(void* tables, int32_t nfft, int32_t nperseg, int32_t window_id, int32_t bin_lo,
                                     int32_t bin_hi, lbm_stream_t stream) {
    LBM_REQUIRE(tables != nullptr && ((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_welch_tables_init: tables NULL or not 16-byte aligned");
    LBM_REQUIRE(nperseg >= 1 && nperseg <= nfft, LBM_EINVAL, "lbm_welch_tables_init: need 1 <= nperseg <= nfft");
    LBM_REQUIRE(window_id >= LBM_WIN_BOXCAR && window_id <= LBM_WIN_HAMMING, LBM_EINVAL, "lbm_welch_tables_init: unknown window_id %d", window_id);
    LBM_REQUIRE(bin_lo >= 0 && bin_lo <= bin_hi && bin_hi <= nfft / 2, LBM_EINVAL, "lbm_welch_tables_init: bins [%d,%d] outside [0,%d]", bin_lo, bin_hi, nfft / 2);
    hipLaunchKernelGGL(welch_tables_kernel, dim3(1), dim3(kThreads), 0, lbm::as_stream(stream), tables, nfft, nperseg, window_id,
                       bin_lo, bin_hi);
    return lbm::check_launch("lbm_welch_tables_init");
}

namespace {
// keep != nullptr: the caller wants the reject test fused; *keep_done says whether this launch did it (team path only)
int welch_impl(const float* x, const float* row_scale, float* psd, float* seg_or_null, int64_t rows, int32_t T, float fs,
               int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
               int32_t bin_hi, int32_t average_mode, lbm_stream_t stream, uint8_t* keep = nullptr, int C = 1, float ptp_thresh = 0.f,
               bool* keep_done = nullptr) {
    LBM_REQUIRE(rows >= 0, LBM_EINVAL, "lbm_welch_f32: negative rows");
    LBM_REQUIRE(nperseg >= 1 && nperseg <= nfft, LBM_EINVAL, "lbm_welch_f32: need 1 <= nperseg <= nfft");
    LBM_REQUIRE(nfft <= T, LBM_EINVAL, "lbm_welch_f32: n_fft (%d) is larger than the number of samples (%d)", nfft, T);
    LBM_REQUIRE(noverlap >= 0 && noverlap < nperseg, LBM_EINVAL, "lbm_welch_f32: need 0 <= noverlap < nperseg");
    LBM_REQUIRE(bin_lo >= 0 && bin_lo <= bin_hi && bin_hi <= nfft / 2, LBM_EINVAL, "lbm_welch_f32: bins [%d,%d] outside [0,%d]", bin_lo, bin_hi, nfft / 2);
    LBM_REQUIRE(average_mode >= LBM_AVG_MEAN && average_mode <= LBM_AVG_NONE, LBM_EINVAL, "lbm_welch_f32: unknown average_mode %d", average_mode);
    LBM_REQUIRE(fs > 0.f, LBM_EINVAL, "lbm_welch_f32: fs must be positive");
    if (rows == 0) return LBM_OK;
    LBM_REQUIRE(x != nullptr && tables != nullptr, LBM_EINVAL, "lbm_welch_f32: NULL pointer");
    LBM_REQUIRE(average_mode == LBM_AVG_NONE ? seg_or_null != nullptr : psd != nullptr, LBM_EINVAL, "lbm_welch_f32: NULL output");
    LBM_REQUIRE(rows <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_welch_f32: too many rows");
    const int hop = nperseg - noverlap;
    const int nseg = (T - noverlap) / hop;
    const int F = bin_hi - bin_lo + 1;
    // fast path: folded pruned DFT on MFMA
    const int n_even = (bin_hi / 2) - ((bin_lo + 1) / 2) + 1, n_odd = F - n_even;
    if (fold_ok_tables(nfft, nperseg) && average_mode == LBM_AVG_MEAN && seg_or_null == nullptr && n_even <= 16 &&
        n_odd <= 16 && nseg >= 1) {
        const FoldGeom g = fold_geom(nfft, T, hop, nseg);
        const size_t shared_floats = (size_t)16 * g.LDB + ((g.H + 1 + 3) & ~3);
        // team path: 16-byte rows; teams (= workgroups) of 2 waves, of 4 for long rows
        if (((T & 3) == 0) && ((((uintptr_t)x) & 15) == 0) && g.TILES <= 2) {
            const int ts = g.rows_floats <= 6000 ? 2 : 4;
            const size_t lds_t = ((size_t)team_rows_pad(g.rows_floats) + (size_t)g.TILES * (ts - 1) * 1024 +
                                  (size_t)g.TILES * 16 * (kLdp + 1) + ((g.H + 1 + 3) & ~3) + (keep ? (size_t)g.RW * ts * 2 : 0)) * sizeof(float);
            // every step of a wave's straight line must stay inside its segment: n <= H for the last wave's last step
            const bool in_seg = 4 * (((ts - 1) * g.QS) / ts + kTeamSteps - 1) + 3 <= g.H;
            if ((g.QS + ts - 1) / ts <= kTeamSteps && g.QS >= 2 * ts && in_seg && lds_t <= 80 * 1024) {
                const int64_t ngroups = (rows + g.RW - 1) / g.RW;
                int per_cu = (int)((160 * 1024) / lds_t);
                const int cap = (ts == 2 && g.TILES == 1) ? 6 : 8 / ts;   // the kernels' register budgets: 3 (2) waves per SIMD
                if (per_cu > cap) per_cu = cap;
                int64_t blocks = 256LL * per_cu;
                if (blocks > ngroups) blocks = ngroups;
                auto k = ts == 2 ? (g.TILES == 1 ? welch_team_kernel<2, 1> : welch_team_kernel<2, 2>)
                                 : (g.TILES == 1 ? welch_team_kernel<4, 1> : welch_team_kernel<4, 2>);
                if (keep != nullptr) {
                    k = ts == 2 ? (g.TILES == 1 ? welch_team_kernel<2, 1, true> : welch_team_kernel<2, 2, true>)
                                : (g.TILES == 1 ? welch_team_kernel<4, 1, true> : welch_team_kernel<4, 2, true>);
                    if (keep_done) *keep_done = true;
                }
                if (lds_t > 48 * 1024 &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_t) != hipSuccess)
                    return lbm::fail(LBM_ELAUNCH, "lbm_welch_f32(team): cannot reserve %zu bytes of LDS", lds_t);
                hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(64 * ts), lds_t, lbm::as_stream(stream), x, row_scale, psd, rows, T,
                                   (double)fs, g, tables, bin_lo, F, ngroups, keep, C, ptp_thresh);
                return lbm::check_launch("lbm_welch_f32(team)");
            }
        }
        int waves = 6;   // 2 + 2 + 1 + 1 per SIMD: the phases of a group are chains of LDS round trips, other waves fill them
        while (waves > 1 && (shared_floats + (size_t)waves * g.wave_floats) * sizeof(float) > 160 * 1024) --waves;
        const size_t lds_f = (shared_floats + (size_t)waves * g.wave_floats) * sizeof(float);
        if (lds_f <= 160 * 1024) {
            const int64_t ngroups = (rows + g.RW - 1) / g.RW;
            int64_t blocks = (ngroups + waves - 1) / waves;
            if (blocks > 1024) blocks = 1024;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(welch_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f);
            if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_welch_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(welch_fold_kernel, dim3((unsigned)blocks), dim3(64 * waves), lds_f, lbm::as_stream(stream), x, row_scale, psd, rows, T,
                               (double)fs, g, tables, bin_lo, F, ngroups);
            return lbm::check_launch("lbm_welch_f32(fold)");
        }
    }
    const size_t lds = welch_lds_bytes(nfft, nperseg, nseg, F);
    LBM_REQUIRE(lds <= 160 * 1024, LBM_EUNSUPPORTED, "lbm_welch_f32: row needs %zu B of LDS (max 163840)", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(welch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_welch_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(welch_kernel, dim3((unsigned)rows), dim3(kThreads), lds, lbm::as_stream(stream), x, row_scale, psd, seg_or_null, T,
                       (double)fs, nfft, nperseg, hop, nseg, tables, bin_lo, F, average_mode);
    return lbm::check_launch("lbm_welch_f32");
}
}  // namespace

extern "C" int lbm_welch_f32(const float* x, float* psd, float* seg_or_null, int64_t rows, int32_t T, float fs,
                             int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
                             int32_t bin_hi, int32_t average_mode, lbm_stream_t stream) {
    return welch_impl(x, nullptr, psd, seg_or_null, rows, T, fs, nfft, nperseg, noverlap, tables, bin_lo, bin_hi, average_mode, stream);
}

extern "C" int lbm_welch_scaled_f32(const float* x, const float* row_scale, float* psd, float* seg_or_null, int64_t rows, int32_t T,
                                    float fs, int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
                                    int32_t bin_hi, int32_t average_mode, lbm_stream_t stream) {
    LBM_REQUIRE(rows == 0 || row_scale != nullptr, LBM_EINVAL, "lbm_welch_scaled_f32: NULL row_scale");
    return welch_impl(x, row_scale, psd, seg_or_null, rows, T, fs, nfft, nperseg, noverlap, tables, bin_lo, bin_hi, average_mode, stream);
}

// post.hip
extern "C" int lbm_epoch_post_f32(const float* x, float* y_base, float* y_std, uint8_t* keep, int64_t E, int32_t C, int32_t T,
                                  int32_t n_baseline, float ptp_thresh, int32_t standardize_mode, float eps, lbm_stream_t stream);

extern "C" int lbm_welch_keep_f32(const float* x, float* psd, uint8_t* keep, int64_t E, int32_t C, int32_t T, float fs, int32_t nfft,
                                  int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo, int32_t bin_hi, float ptp_thresh,
                                  lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1, LBM_EINVAL, "lbm_welch_keep_f32: bad sizes");
    if (E == 0) return LBM_OK;
    LBM_REQUIRE(keep != nullptr, LBM_EINVAL, "lbm_welch_keep_f32: NULL keep");
    hipError_t e = hipMemsetAsync(keep, 1, (size_t)E, lbm::as_stream(stream));
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_welch_keep_f32: memset keep: %s", hipGetErrorString(e));
    bool fused = false;
    const int rc = welch_impl(x, nullptr, psd, nullptr, E * C, T, fs, nfft, nperseg, noverlap, tables, bin_lo, bin_hi, LBM_AVG_MEAN, stream,
                              keep, C, ptp_thresh, &fused);
    if (rc != LBM_OK || fused) return rc;
    // the launch above was not the team kernel: the reject test as its own pass (the post kernel in its mask-only form)
    return lbm_epoch_post_f32(x, nullptr, nullptr, keep, E, C, T, 0, ptp_thresh, LBM_STD_NONE, 0.f, stream);
}

extern "C" int lbm_bandpower_f32(const float* psd, const float* freqs, float* out, int64_t E, int32_t C, int32_t F,
                                 int32_t per_channel, int32_t already_db, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && F >= 1, LBM_EINVAL, "lbm_bandpower_f32: bad sizes");
    if (E == 0) return LBM_OK;
    LBM_REQUIRE(psd && freqs && out, LBM_EINVAL, "lbm_bandpower_f32: NULL pointer");
    const int64_t groups = per_channel ? E * C : E;
    const int n_elem = per_channel ? F : C * F;
    const int64_t blocks = (groups + 3) / 4;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_bandpower_f32: grid too large");
    hipLaunchKernelGGL(bandpower_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), psd, freqs, out, groups,
                       n_elem, F, already_db);
    return lbm::check_launch("lbm_bandpower_f32");
}
