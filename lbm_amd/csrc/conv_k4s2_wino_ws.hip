// lbm_amd -- the encoder's 4-tap stride-2 layers (Conv1d 30 -> 64 and 64 -> 128, kernel 4, stride 2, padding 1 + ReLU; reference:
// training/model.py:169-172) as a WEIGHTS-STATIONARY minimal-filtering kernel (round 3).
//
// y[p] = sum_c  w[c][0] x[c][2p-1] + w[c][1] x[c][2p] + w[c][2] x[c][2p+1] + w[c][3] x[c][2p+2]
//      = sum_c  (w1 xe[p] + w3 xe[p+1])  +  (w0 xo[p-1] + w2 xo[p]),      xe[q] = x[2q], xo[q] = x[2q+1]:
// two 2-tap stride-1 correlations per input channel, one over the even and one over the odd samples.  Each is computed three outputs at a
// time by the minimal form F(3,2) -- 4 products per 3 outputs instead of 6 (points 0, 1, -1, inf; the same input transform as F(2,3)):
//   V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3        (d0..d3 = four consecutive samples of the phase)
//   U0 = g0        U1 = (g0 + g1)/2   U2 = (g0 - g1)/2   U3 = g1     (g0, g1 = the phase's two taps)
//   m_i = sum over (channel, phase) of U_i V_i
//   y0 = m0 + m1 + m2   y1 = m1 - m2   y2 = m1 + m2 - m3
// i.e. four GEMMs with K = 2 Cin over output TRIPLES: 2/3 of the direct form's matrix work.  Not bit-identical to the direct kernels
// (conv_k4s2_ws_kernel, conv_tile_kernel): the transforms round differently; measured against float64 it is as close as they are
// (tests: test_conv1d_k4s2_wino_matches_float64).
//
// Work layout = conv_wino3_ws_kernel's (conv_wino_ws.hip): a WAVE owns one transform i and 32 output channels for the whole launch, its
// U_i[32][2 Cin] is the MFMA B operand in registers; persistent 8-wave workgroups (4 transforms x 2 channel groups, one per CU) walk
// tiles of 64 output triples (192 positions); only the raw input window moves -- Cin / 2 input rows x 392 samples (25 | 49 KB) per phase,
// two phases per tile, by LDS-DMA into one of two buffers, requested under the previous phase's MFMAs; the four transforms of an output
// meet in a 48 KB LDS exchange area under the next tile's MFMAs (two accumulator sets; the register quad a wave outputs itself never
// leaves its registers, which is why there is one instance of the tile walk per transform: operand picks, exchange slots and the own
// quad are compile-time).  The previous tile's stores are issued well behind the window requests (hook 5 of 8 / 2 of 4: next to them
// they queue in front of the requests; measurements in profiles/r03_experiments.md).  Row r of position sub-tile pm is triple 2 r + pm, so a lane's two triples are
// neighbours: the 14 samples W0..W13 = x[6 (2 r) - 1 ...] they need from one input row sit in three aligned 16-byte quads (+ one
// sample for transforms 0 and 3), read once for 4 MFMAs (2 phases x 2 sub-tiles):
//   even phase  sub-tile 0: d = W1 W3 W5 W7     sub-tile 1: d = W7 W9 W11 W13
//   odd  phase  sub-tile 0: d = W0 W2 W4 W6     sub-tile 1: d = W6 W8 W10 W12
// An MFMA k-step is (input rows 2 u, 2 u + 1 in the two lane halves) x (one phase).
//
// Requires Cin <= 64 (rows beyond Cin carry zero weights and are copied from a zero source), Cout a multiple of 64, 16-byte rows.
#include <type_traits>

#include "common.h"
#include "conv_ws.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kXW = 392;                        // window samples per input row: x[384 pt - 4 .. 384 pt + 387]
constexpr int kX4 = kXW / 4;                    // 98 quads
constexpr int kExFloats = 8 * 2 * 12 * 64;      // [wave][position sub-tile][3 register quads][lane]: a wave keeps the quad it outputs itself

template <int CINP>
struct Geo {                                    // a tile's window is staged in two phases of CINP / 2 input rows
    static constexpr int kPhRows = CINP / 2;                  // 16 | 32
    static constexpr int kPhQuads = kPhRows * kX4;            // 1 568 | 3 136
    static constexpr int kPieces = (kPhQuads + 63) / 64;      // 25 (the last one half used) | 49 DMA pieces of 1 KiB
    static constexpr int kBufFloats = kPieces * 256;          // 25 600 | 50 176 B per buffer
    static constexpr int kSlots = (kPieces + 7) / 8;          // pieces per wave: 4 | 7 (wave 0; the others one fewer)
    static constexpr int kHooks = kPhRows / 4;                // side-work slots per phase: one per two row pairs (8 MFMAs)
    static constexpr size_t kLdsBytes = (2 * (size_t)kBufFloats + kExFloats) * sizeof(float);   // 100 352 | 149 504
    static_assert(kSlots <= kHooks, "two pieces are requested per hook in the first half of a phase");
    static_assert(kHooks >= 4, "the four parts of the exchange are written from hooks 0..3");
};

// LDS-DMA through inline assembly (see conv_wino_ws.hip / vq_screen.hip)
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// Source of the window quads that lie wholly outside [0, Lin) (the zero padding, everything right of a row's end) or in rows >= Cin
__device__ __align__(16) float g_zero_quad_kw[4];

template <int CINP>   // input rows the phases cover: 32 (layer 1: Cin = 30) or 64
__global__ __launch_bounds__(kThreads) void conv_k4s2_wino_ws_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                     const float* __restrict__ bias, float* __restrict__ y, int ldx, int ldy,
                                                                     int Cin, int Lin, int Lout, int Cout, int relu, int ptiles,
                                                                     int64_t ntiles, int ncg, int nstreams, int xcd_map) {
    using G = Geo<CINP>;
    constexpr int kPhRows = G::kPhRows, kPhQuads = G::kPhQuads, kPieces = G::kPieces, kBufFloats = G::kBufFloats, kSlots = G::kSlots;
    constexpr int NPH = 2;
    extern __shared__ __align__(16) float smem[];
    float* const s_x = smem;                          // 2 x [CINP / 2 rows][392]
    float* const s_ex = smem + 2 * kBufFloats;        // exchange area
    const unsigned lds_x = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)s_x;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int ti = wave & 3, c32 = wave >> 2;          // this wave's transform and its 32-channel group inside the workgroup's 64

    int cg, stream;
    if (xcd_map) {                                      // the ncg workgroups of a stream on one XCD: they share the window through its L2
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        cg = r % ncg;
        stream = (r / ncg) * 8 + xcd;
    } else {
        cg = blockIdx.x % ncg;
        stream = blockIdx.x / ncg;
    }
    const int co = cg * 64 + c32 * 32 + j;             // this lane's output channel (MFMA column)

    // ---- B operand: k-step t = 2 u + phase of lane half h is U_ti of input row 2 u + h: even phase taps (w1, w3), odd phase (w0, w2)
    float ureg[CINP];
    {
        const f4* wr = reinterpret_cast<const f4*>(w) + (int64_t)co * Cin;
#pragma unroll
        for (int u = 0; u < CINP / 2; ++u) {
            const int c = 2 * u + h;
            const f4 v = wr[c < Cin ? c : 0];
            float ge0 = v.y, ge1 = v.w, go0 = v.x, go1 = v.z;
            if (c >= Cin) ge0 = ge1 = go0 = go1 = 0.f;
            ureg[2 * u] = ti == 0 ? ge0 : ti == 3 ? ge1 : ti == 1 ? 0.5f * (ge0 + ge1) : 0.5f * (ge0 - ge1);
            ureg[2 * u + 1] = ti == 0 ? go0 : ti == 3 ? go1 : ti == 1 ? 0.5f * (go0 + go1) : 0.5f * (go0 - go1);
        }
    }
    const float bv = bias != nullptr ? bias[co] : 0.f;
    const float lo = relu ? 0.f : -INFINITY;

    // ---- DMA pieces of this wave: piece p = wave + 8 k covers window quads 64 p .. 64 p + 63 (row = quad / 98, column quad = quad % 98)
    // (the piece's first quad is wave-uniform: its row and column come off the scalar unit, a lane adds itself with one carry; no
    // per-piece registers.  Row = kPhRows: past the phase's rows -- the tail of the last piece --: zero source)
    auto piece_rc = [&](int k, int& qr, int& c4) {
        const int A = 64 * (wave + 8 * k);
        c4 = A % kX4 + lane;
        const bool carry = c4 >= kX4;
        c4 -= carry ? kX4 : 0;
        qr = A / kX4 + (carry ? 1 : 0);
    };
    const int npieces = wave < kPieces - 8 * (kSlots - 1) ? kSlots : kSlots - 1;
    // the zero source's address lives in a register pair: fetched per request it costs a scalar load and a wait that also drains the LDS queue
    const float* vzero = g_zero_quad_kw;
    asm volatile("" : "+v"(vzero));

    auto request = [&](int64_t b, int pt, int ph, int buf, int k) {
        if (k >= npieces) return;                      // wave-uniform
#if defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 2)
        if (b != 0) return;   // (ablation: no window copies after the first epoch)
#endif
#if defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 32)
        // (ablation: the same number of copies, no address arithmetic: every piece copies the first KiB of the epoch)
        { lds_dma16(x + (b * Cin * (int64_t)ldx + 4 * lane), lds_x + (unsigned)(buf * kBufFloats + 256 * (wave + 8 * k)) * 4u); return; }
#endif
        int qr, c4;
        piece_rc(k, qr, c4);
        // wave-uniform 64-bit base (scalar unit) + a 32-bit lane offset (one 24-bit multiply-add); branch-free choice of the source
        const int row0 = ph * kPhRows;
        const float* base = x + (b * Cin + row0) * (int64_t)ldx;
        const int g = pt * 384 - 4 + 4 * c4;
        const bool ok = qr < kPhRows && row0 + qr < Cin && (unsigned)g < (unsigned)Lin;
        const unsigned off = __umul24((unsigned)qr, (unsigned)ldx) + (unsigned)g;      // qr <= 32, ldx < 2^24 (launcher)
        const float* src = base + off;
        const unsigned long long sa = (unsigned long long)src, za = (unsigned long long)vzero;
        const unsigned lo32 = ok ? (unsigned)sa : (unsigned)za, hi32 = ok ? (unsigned)(sa >> 32) : (unsigned)(za >> 32);
        lds_dma16(reinterpret_cast<const float*>(((unsigned long long)hi32 << 32) | lo32),
                  lds_x + (unsigned)(buf * kBufFloats + 256 * (wave + 8 * k)) * 4u);
    };
    // After this wave's pieces have landed: the quad that straddles a row's end (Lin not a multiple of 4) holds pitch padding behind the
    // last real sample; the lane that copied it zeroes that.  Wave-uniform test first.
    auto fix_edges = [&](int pt, int buf) {
        const int ga = pt * 384 - 4;
        if ((Lin & 3) == 0 || ga + kXW <= Lin || ga >= Lin) return;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            if (k < npieces) {
                int qr, c4;
                piece_rc(k, qr, c4);
                const int g = ga + 4 * c4;
                if (qr < kPhRows && g < Lin && g + 4 > Lin) {
                    float* d = s_x + buf * kBufFloats + 256 * (wave + 8 * k) + 4 * lane;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (g + e >= Lin) d[e] = 0.f;
                }
            }
        }
    };

    f32x16 acc[2][2];   // [tile parity][position sub-tile]
    // One phase = 8 input-row pairs = 32 MFMAs.  Window slot s of a row = sample 384 pt - 4 + s; W_k of row step j sits at slot 12 j + 3 + k:
    // quads A = W1..W4 (slot 12 j + 4), B = W5..W8, C = W9..W12, and the single samples W0 (slot 12 j + 3) and W13 (slot 12 j + 16).
    auto mfma_phase = [&](auto ti_tag, auto par_tag, auto ph_tag, int buf, auto&& hook) {
        constexpr int TI = decltype(ti_tag)::value, P = decltype(par_tag)::value, PH = decltype(ph_tag)::value;
        constexpr int KIND = TI == 0 ? 0 : TI == 3 ? 3 : 1;
        typedef const volatile __attribute__((address_space(3))) f4* lds_vf4;
        typedef const volatile __attribute__((address_space(3))) float* lds_vf1;
        const lds_vf4 xq = (lds_vf4)(s_x + buf * kBufFloats + h * kXW + 12 * j + 4);
        const lds_vf1 xe = (lds_vf1)(s_x + buf * kBufFloats + h * kXW + 12 * j + (KIND == 0 ? 3 : 16));
        constexpr int NU = kPhRows / 2, RD = 1;   // sub-step = one row pair (4 MFMAs); the reads run RD sub-steps ahead of the MFMAs
        f4 tq[RD + 1][3];
        float te[RD + 1] = {};
        float v[2][4];
        auto fetch = [&](int u, int set) {
#if defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 4)
            if (u > 0) { tq[set][0] = tq[0][0] + 1.f; tq[set][1] = tq[0][1]; tq[set][2] = tq[0][2]; te[set] = te[0]; return; }   // (ablation: one operand read per phase)
#endif
            const int off = 2 * u * kXW;
            tq[set][0] = xq[off / 4];
            tq[set][1] = xq[off / 4 + 1];
            tq[set][2] = xq[off / 4 + 2];
            if (KIND != 1) te[set] = xe[off];
        };
        // operands of sub-step u: [0] even phase / sub-tile 0, [1] even / 1, [2] odd / 0, [3] odd / 1 -- each in its own register
        auto form = [&](int u, int part) {
            const f4 A = tq[u % (RD + 1)][0], B = tq[u % (RD + 1)][1], C = tq[u % (RD + 1)][2];
            const float e = te[u % (RD + 1)];
            float a, b;
            if (part == 0) {
                if (KIND == 1) { a = TI == 1 ? B.x + A.z : B.x - A.z; b = TI == 1 ? C.z + C.x : C.z - C.x; }   // d2 +- d1: W5 W3 | W11 W9
                else if (KIND == 0) { a = A.x - B.x; b = B.z - C.z; }                     // d0 - d2 = W1 - W5 | W7 - W11
                else { a = A.z - B.z; b = C.x - e; }                                      // d1 - d3 = W3 - W7 | W9 - W13
            } else {
                if (KIND == 1) { a = TI == 1 ? A.w + A.y : A.w - A.y; b = TI == 1 ? C.y + B.w : C.y - B.w; }   // W4 W2 | W10 W8
                else if (KIND == 0) { a = e - A.w; b = B.y - C.y; }                       // W0 - W4 | W6 - W10
                else { a = A.y - B.y; b = B.w - C.w; }                                    // W2 - W6 | W8 - W12
            }
            asm volatile("" : "+v"(a), "+v"(b));
            v[u & 1][2 * part] = a; v[u & 1][2 * part + 1] = b;
        };
#pragma unroll
        for (int u = 0; u < RD; ++u) fetch(u, u);
        form(0, 0);
        form(0, 1);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + RD < NU) fetch(u + RD, (u + RD) % (RD + 1));
            __builtin_amdgcn_sched_barrier(0);
            const int t = PH * kPhRows + 2 * u;
            if (PH == 0 && u == 0) {
                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][0], ureg[t], z, 0, 0, 0);
                acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][1], ureg[t], z, 0, 0, 0);
            } else {
                acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][0], ureg[t], acc[P][0], 0, 0, 0);
                acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][1], ureg[t], acc[P][1], 0, 0, 0);
            }
            if (u + 1 < NU) form(u + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][2], ureg[t + 1], acc[P][0], 0, 0, 0);
            acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][3], ureg[t + 1], acc[P][1], 0, 0, 0);
            if (u + 1 < NU) form(u + 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            if (u & 1) hook(u >> 1);   // (kHooks per phase) side work while these MFMAs occupy the pipe: window requests, the previous tile's exchange and outputs
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // exchange area: [wave][pm][slot][lane] float4 -- register quad rq of a lane = rows 8 rq + 4 h + (0..3) of column j; the wave of
    // transform TI outputs quad TI and keeps its own share of it in registers, its other three quads go to slots 0..2 in order.
    auto ex_write = [&](auto ti_tag, auto par_tag, int q) {
        constexpr int TI = decltype(ti_tag)::value, P = decltype(par_tag)::value;
        const int pm = q >> 1;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int rq = 2 * (q & 1) + r2;
            if (rq == TI) continue;
            f4 vv = {acc[P][pm][4 * rq], acc[P][pm][4 * rq + 1], acc[P][pm][4 * rq + 2], acc[P][pm][4 * rq + 3]};
            reinterpret_cast<f4*>(s_ex)[((wave * 2 + pm) * 3 + (rq < TI ? rq : rq - 1)) * 64 + lane] = vv;
        }
    };
    // wave (ti, c32) forms register quad ti of channel group c32.  Row r = 8 ti + 4 h + e holds triple 2 r + pm: the two sub-tiles of a row are
    // six consecutive outputs, a lane's four rows 24 of them: six 16-byte stores, 96 contiguous bytes per lane.
    auto ex_output = [&](auto ti_tag, auto par_tag, int64_t b, int pt, int eh_lo, int eh_hi) {
        constexpr int TI = decltype(ti_tag)::value, P = decltype(par_tag)::value;   // P: the accumulator set of the tile being output
        // (row base: the wave-uniform part in 64 bits on the scalar unit, the lane's channel as one 24-bit multiply)
        float* yr = y + (b * Cout + (cg * 64 + c32 * 32)) * (int64_t)ldy + __umul24((unsigned)j, (unsigned)ldy);
        const int p0 = pt * 192 + 48 * ti + 24 * h;
        typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int eh = eh_lo; eh < eh_hi; ++eh) {   // rows e = 2 eh, 2 eh + 1 at a time (12 outputs, three stores): 16 + 12 registers live, not 32 + 24
            f2 m[2][4];
#pragma unroll
            for (int pm = 0; pm < 2; ++pm)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k == TI) m[pm][k] = (f2){acc[P][pm][4 * TI + 2 * eh], acc[P][pm][4 * TI + 2 * eh + 1]};
                    else m[pm][k] = reinterpret_cast<const f2*>(s_ex)[((((k + 4 * c32) * 2 + pm) * 3 + (TI < k ? TI : TI - 1)) * 64 + lane) * 2 + eh];
                }
            float o[12];
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int pm = 0; pm < 2; ++pm) {
                    float m0 = m[pm][0][e], m1 = m[pm][1][e], m2 = m[pm][2][e], m3 = m[pm][3][e];
                    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3));   // scalars: hipcc otherwise pairs the adds of e = 0, 1 into
                                                                                  // v_pk_add_f32, which costs more beside MFMAs than the pair
                    o[6 * e + 3 * pm] = fmaxf(((m0 + m1) + m2) + bv, lo);
                    o[6 * e + 3 * pm + 1] = fmaxf((m1 - m2) + bv, lo);
                    o[6 * e + 3 * pm + 2] = fmaxf(((m1 + m2) - m3) + bv, lo);
                }
#pragma unroll
            for (int s3 = 0; s3 < 3; ++s3) {
                const int p = p0 + 12 * eh + 4 * s3;
                // ldy is a multiple of 4 >= Lout: the tail quad may spill into the row's pitch padding (finite values)
#if defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 1)
                if (o[4 * s3] + o[4 * s3 + 1] + o[4 * s3 + 2] + o[4 * s3 + 3] == 123.456f) yr[p] = 0.f;   // (ablation: no stores)
#elif defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 16)
                // (ablation: the same number of 16-byte stores, but a wave's 64 lanes write 1 KiB of consecutive addresses -- wrong places)
                if (p < Lout) *reinterpret_cast<f4*>(y + ((b * Cout + cg * 64 + wave * 8) * (int64_t)ldy + pt * 192 + (eh * 3 + s3) * 256 + 4 * lane) % ((int64_t)Cout * ldy * 16384 - 1024) / 4 * 4) = (f4){o[4 * s3], o[4 * s3 + 1], o[4 * s3 + 2], o[4 * s3 + 3]};
#else
                if (p < Lout) *reinterpret_cast<f4*>(yr + p) = (f4){o[4 * s3], o[4 * s3 + 1], o[4 * s3 + 2], o[4 * s3 + 3]};
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- the tile walk: tile = stream, stream + nstreams, ...; (b, pt) kept incrementally
    int64_t tile = stream;
    if (tile >= ntiles) return;                        // workgroup-uniform
    int64_t b = tile / ptiles;
    int pt = (int)(tile - b * ptiles);
    const int db = nstreams / ptiles, dpt = nstreams - db * ptiles;
    int64_t pb = 0;
    int ppt = 0;
    bool have_prev = false;

#pragma unroll
    for (int k = 0; k < kSlots; ++k) request(b, pt, 0, 0, k);
    // one tile into accumulator set P; returns true after the last tile of the walk (whose exchange it then completes)
    auto tile_step = [&](auto ti_tag, auto par_tag) -> bool {
        constexpr int P = decltype(par_tag)::value;
        const std::integral_constant<int, 1 - P> other{};
        int64_t nb = b + db;
        int npt = pt + dpt;
        if (npt >= ptiles) { npt -= ptiles; ++nb; }
        const int64_t ntile = tile + nstreams;
        const bool more = ntile < ntiles;
        auto phase = [&](auto ph_tag) {
            constexpr int PH = decltype(ph_tag)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fix_edges(pt, PH & 1);
            __syncthreads();   // this phase's rows are in buffer PH & 1; every wave is done with the other buffer (and, in phase 1, has
                               // written the previous tile's transforms; in phase 0, has read the outputs of the tile before that)
            mfma_phase(ti_tag, par_tag, ph_tag, PH & 1, [&](int g) {
// (measured, profiles/r03_experiments.md: the previous tile's stores issued well BEHIND the window requests -- hook 5 of 8 / 2 of 4 --
// instead of next to them: layer 2 3.14 -> 2.95 ms, layer 1 2.05 -> 1.85 ms; at the last hooks they are not done by the phase's end)
#ifndef LBM_KW_OUT_HOOK
#define LBM_KW_OUT_HOOK (5 * G::kHooks / 8)
#endif
#ifndef LBM_KW_OUT_HOOK_B
#define LBM_KW_OUT_HOOK_B LBM_KW_OUT_HOOK
#endif
#ifndef LBM_KW_EXW_HOOK0
#define LBM_KW_EXW_HOOK0 0
#endif
#ifndef LBM_KW_REQ_HOOK0
#define LBM_KW_REQ_HOOK0 0
#endif
                const int gr = g - LBM_KW_REQ_HOOK0;
                if (gr >= 0 && 2 * gr < kSlots) {   // the next phase's window (the next tile's first one in the last phase) into the other buffer
                    if (PH + 1 < NPH) { request(b, pt, PH + 1, (PH + 1) & 1, 2 * gr); request(b, pt, PH + 1, (PH + 1) & 1, 2 * gr + 1); }
                    else if (more) { request(nb, npt, 0, 0, 2 * gr); request(nb, npt, 0, 0, 2 * gr + 1); }
                }
#if defined(LBM_KW_ABLATE) && (LBM_KW_ABLATE & 8)
                if (PH == 1 && g == 0 && have_prev && bv == 123.456f) { ex_write(ti_tag, other, 0); ex_output(ti_tag, other, pb, ppt, 0, 2); }   // (ablation: no exchange)
#else
                if (PH == 0 && g >= LBM_KW_EXW_HOOK0 && g < LBM_KW_EXW_HOOK0 + 4 && have_prev) ex_write(ti_tag, other, g - LBM_KW_EXW_HOOK0);
                if (PH == 1 && g == LBM_KW_OUT_HOOK && have_prev) ex_output(ti_tag, other, pb, ppt, 0, 1);
                if (PH == 1 && g == LBM_KW_OUT_HOOK_B && have_prev) ex_output(ti_tag, other, pb, ppt, 1, 2);
#endif
            });
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        pb = b; ppt = pt; have_prev = true;
        if (more) {
            tile = ntile; b = nb; pt = npt;
            return false;
        }
        __syncthreads();           // the exchange area is free again
#pragma unroll
        for (int q = 0; q < 4; ++q) ex_write(ti_tag, par_tag, q);
        __syncthreads();
        ex_output(ti_tag, par_tag, pb, ppt, 0, 2);
        return true;
    };
    auto walk = [&](auto ti_tag) {
        while (true) {
            if (tile_step(ti_tag, std::integral_constant<int, 0>{})) break;
            if (tile_step(ti_tag, std::integral_constant<int, 1>{})) break;
        }
    };
    // wave-uniform: one instance of the walk per transform (operand reads, exchange slots and the output quad are compile-time)
    if (ti == 0) walk(std::integral_constant<int, 0>{});
    else if (ti == 1) walk(std::integral_constant<int, 1>{});
    else if (ti == 2) walk(std::integral_constant<int, 2>{});
    else walk(std::integral_constant<int, 3>{});
}

int g_cus_kw[64];   // multiprocessor count per device (0 = not asked yet)

}  // namespace

namespace lbm {

bool k4s2_wino_eligible(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, int Cin, int Cout, int Lin) {
    return Cin >= 1 && Cin <= 64 && Cout % 64 == 0 && Cout >= 64 && Lin >= 2 && ldx % 4 == 0 && ldy % 4 == 0 && ldx < (1 << 24) && ldy < (1 << 24) &&
           ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w)) & 15) == 0;
}

int launch_k4s2_wino(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout, int Lout,
                     int64_t ldy, int relu, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(LBM_ELAUNCH, "lbm_conv1d_k4s2_wino_f32: hipGetDevice failed");
    if (g_cus_kw[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k4s2_wino_ws_kernel<32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)Geo<32>::kLdsBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_k4s2_wino_ws_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)Geo<64>::kLdsBytes);
        if (e != hipSuccess) return fail(LBM_ELAUNCH, "lbm_conv1d_k4s2_wino_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        g_cus_kw[dev] = n;
    }
    const int ntr = (Lout + 2) / 3;                 // output triples per row
    const int ptiles = (ntr + 63) / 64;
    const int64_t ntiles = B * ptiles;
    const int ncg = Cout / 64;
    int nstreams = g_cus_kw[dev] / ncg;
    if (nstreams < 1) nstreams = 1;
    int xcd_map = 0;
    if (nstreams >= 8) {
        nstreams &= ~7;
        xcd_map = 1;
    }
    if ((int64_t)nstreams > ntiles) {
        nstreams = (int)ntiles;
        xcd_map = 0;
    }
    auto k = Cin <= 32 ? conv_k4s2_wino_ws_kernel<32> : conv_k4s2_wino_ws_kernel<64>;
    hipLaunchKernelGGL(k, dim3((unsigned)(nstreams * ncg)), dim3(kThreads), Cin <= 32 ? Geo<32>::kLdsBytes : Geo<64>::kLdsBytes, st, x, w, bias, y, (int)ldx, (int)ldy, Cin, Lin, Lout,
                       Cout, relu, ptiles, ntiles, ncg, nstreams, xcd_map);
    return check_launch("lbm_conv1d_k4s2_wino_f32");
}

}  // namespace lbm
