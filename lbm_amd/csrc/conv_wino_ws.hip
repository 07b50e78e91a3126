// lbm_amd -- third encoder layer (Conv1d 128 -> 256, 3 taps, stride 1, padding 1 + ReLU; reference: training/model.py:173-174)
// as a WEIGHTS-STATIONARY Winograd F(2,3) kernel (round 3).
//
// Same arithmetic as conv_wino3_kernel<4, true> in conv.hip -- the same four transforms, the same table of transformed weights, the same
// order of products in every accumulator chain, the same output transform -- so the results are that kernel's BIT FOR BIT
// (tests: test_conv1d_wino3_ws_is_the_tile_kernel_bit_for_bit).  What differs is who holds what:
//
//   conv_wino3_kernel      a workgroup owns a 128-position x 128-channel tile, stages a slab of U (24 KB) and a slab of input (4.5 KB) per 8
//                          input channels through registers into LDS, one barrier per 32 MFMAs of a wave: 29 KB staged per 128 MFMAs.
//   this kernel            a WAVE owns ONE transform i and 32 output channels for the whole launch: its U_i[32 ch][128 ci] is its MFMA B
//                          operand in 64 registers, loaded once.  Workgroups (8 waves = 4 transforms x 2 channel groups = 64 channels, one
//                          per CU) are persistent and walk a list of 128-position tiles; only the raw input window moves: 64 input
//                          channels x 140 samples (35 KB) per half tile, dropped into LDS by LDS-DMA (no staging registers, no ds_write,
//                          no arithmetic), requested under the previous half's MFMA stream.  72 KB staged per 1 024 MFMAs.
//                          The four transforms of an output meet in LDS once per tile (64 KB exchange area): wave (i, g) then forms
//                          register quad i of the outputs of channel group g and stores them, 64 contiguous bytes per lane.  With two
//                          accumulator sets the exchange of tile t rides under the MFMA stream of tile t + 1.  Two barriers per tile.
//
// Per k-step a wave makes ONE aligned 16-byte LDS read (plus an 8-byte one for transforms 0 and 3) that carries the raw samples of both
// its position sub-tiles, forms V_i = a + sgn b (V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3; one exact operation each) and
// issues two MFMAs.  Measured on the way (profiles/r03_experiments.md): the first version (4-byte reads, exchange and requests between the
// MFMA phases, edge samples zeroed by branches) ran at the tile kernel's speed; what moved it were, in order, the edge quads copied from a
// zero source instead of being fixed up, side work issued inside the MFMA stream, the operands of a sub-step formed in separate registers
// ahead of its MFMAs, and the neighbour-pair row order that halves the LDS instructions.
//
// Requires Cin = 128, Cout a multiple of 64, 16-byte rows (ldx, ldy multiples of 4, aligned bases); everything else runs conv_wino3_kernel.
#include <type_traits>

#include "common.h"
#include "conv_ws.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f4 __attribute__((ext_vector_type(4)));

#ifndef LBM_WS_RD
#define LBM_WS_RD 2
#endif
#ifndef LBM_WS_SUB
#define LBM_WS_SUB 2
#endif
constexpr int kThreads = 512;
constexpr int kCin = 128;
constexpr int kXW = 140;                       // window samples per input-channel row: positions p0 - 4 .. p0 + 135
constexpr int kX4 = kXW / 4;                   // 35 quads
constexpr int kHalfRows = 64;                  // input channels per half tile
constexpr int kHalfFloats = kHalfRows * kXW;   // 8 960 floats = 35 840 B = exactly 35 DMA pieces of 1 KiB
constexpr int kPieces = kHalfFloats / 256;
constexpr int kSlots = (kPieces + 7) / 8;      // pieces per wave (waves 0..2 own five, the others four)
constexpr int kExFloats = 8 * 2 * 16 * 64;     // [wave][position sub-tile][register][lane]
constexpr size_t kLdsBytes = (2 * (size_t)kHalfFloats + kExFloats) * sizeof(float);   // 137 216

static_assert(kSlots == 5, "the request schedule in the tile walk names five pieces per wave");
static_assert(kPieces * 256 == kHalfFloats, "a half window must be a whole number of 1 KiB pieces");

// LDS-DMA through inline assembly (see vq_screen.hip: with the builtin the compiler stops counting its LDS waits); per-lane 64-bit source
// address, wave-uniform LDS destination through M0 (saved and restored).
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    // (readfirstlane: under scalar-register pressure hipcc has handed this wave-uniform value to the "s" operand in a VGPR)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// Source of the window quads that lie wholly outside [0, Lin) (the convolution's zero padding and everything right of a row's end)
__device__ __align__(16) float g_zero_quad[4];

#ifdef LBM_WS_STAMPS
__device__ unsigned long long* g_ws_stamps = nullptr;   // [block][wave][12]: cycles per phase, summed over the tiles of the walk
#define WS_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_d[i] += t_ - st_t; st_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

__global__ __launch_bounds__(kThreads) void conv_wino3_ws_kernel(const float* __restrict__ x, const float* __restrict__ utab,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int ldx, int ldy,
                                                                 int Lin, int Cout, int relu, int ptiles, int64_t ntiles, int mtiles,
                                                                 int ncg, int nstreams, int xcd_map) {
    extern __shared__ __align__(16) float smem[];
    float* const s_x = smem;                          // 2 x [64 rows][140]
    float* const s_ex = smem + 2 * kHalfFloats;       // exchange area
    const unsigned lds_x = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)s_x;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int ti = wave & 3, c32 = wave >> 2;          // this wave's transform and its 32-channel group inside the workgroup's 64

    // workgroup -> (channel group of 64, tile stream).  With xcd_map the ncg workgroups of a stream sit on one XCD (blocks are dealt to the
    // eight XCDs round robin), so the window they all read comes from that XCD's L2 after its first reader.
    int cg, stream;
    if (xcd_map) {
        const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
        cg = r % ncg;
        stream = (r / ncg) * 8 + xcd;
    } else {
        cg = blockIdx.x % ncg;
        stream = blockIdx.x / ncg;
    }
    const int co = cg * 64 + c32 * 32 + j;             // this lane's output channel (MFMA column)

    // ---- B operand: U_ti[co][ci] for k-step t = 4 ch + c of lane half h is ci = 8 ch + 4 h + c (conv_wino3_kernel's order)
    float ureg[kCin / 2];
    {
        const int mt = co >> 7, col = co & 127;
        const f4* ut = reinterpret_cast<const f4*>(utab) + ((int64_t)mt * 4 + ti) * 256 + 2 * col + h;
#pragma unroll
        for (int ch = 0; ch < kCin / 8; ++ch) {
            const f4 v = ut[(int64_t)ch * mtiles * 4 * 256];
            ureg[4 * ch] = v.x; ureg[4 * ch + 1] = v.y; ureg[4 * ch + 2] = v.z; ureg[4 * ch + 3] = v.w;
        }
    }
    const float bv = bias != nullptr ? bias[co] : 0.f;
    const float lo = relu ? 0.f : -INFINITY;

    // V_1 = d2 + d1, V_2 = d2 - d1 share one code path (window slot 4 = position p0 of the tile; the reads are in mfma_half)
    const float sgn = ti == 1 ? 1.f : -1.f;

    // ---- DMA pieces of this wave: piece p = wave + 8 k covers window quads 64 p .. 64 p + 63 (row = quad / 35, column quad = quad % 35)
    int q_row[kSlots], q_c4[kSlots];
#pragma unroll
    for (int k = 0; k < kSlots; ++k) {
        const int q = 64 * (wave + 8 * k) + lane;
        q_row[k] = q / kX4;
        q_c4[k] = q - q_row[k] * kX4;
    }
    const int npieces = wave < kPieces - 8 * (kSlots - 1) ? kSlots : kSlots - 1;

    // Requests piece k of this wave for half `half` of tile (b, pt) into buffer `buf`.  A quad wholly outside [0, Lin) -- the zero padding
    // left of a row, everything right of its end -- is copied from g_zero_quad instead, so it needs no fixing afterwards.
    // (the zero source's address in a register pair: fetched per request it is a scalar load through the GOT + an `lgkmcnt(0)` wait)
    const float* vzero = g_zero_quad;
    asm volatile("" : "+v"(vzero));
    auto request = [&](int64_t b, int pt, int half, int buf, int k) {
#if defined(LBM_WS_ABLATE) && (LBM_WS_ABLATE & 2)
        if (b != 0) return;   // (ablation: no window copies after the first epoch)
#endif
        if (k >= npieces) return;                      // wave-uniform
        const float* src = x + (b * kCin + half * kHalfRows) * (int64_t)ldx;
        const int g = pt * 128 - 4 + 4 * q_c4[k];
        const float* a = (g < 0 || g >= Lin) ? vzero : src + (q_row[k] * ldx + g);
        lds_dma16(a, lds_x + (unsigned)(buf * kHalfFloats + 256 * (wave + 8 * k)) * 4u);
    };
    // After this wave's pieces have landed: the one quad per row that straddles the row's end (Lin not a multiple of 4) still holds
    // pitch padding behind its last real sample; the lane that copied it zeroes that.  Wave-uniform test first.
    auto fix_edges = [&](int pt, int buf) {
        const int ga = pt * 128 - 4;
        if ((Lin & 3) == 0 || ga + kXW <= Lin || ga >= Lin) return;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            if (k < npieces) {
                const int g = ga + 4 * q_c4[k];
                if (g < Lin && g + 4 > Lin) {
                    float* d = s_x + buf * kHalfFloats + 256 * (wave + 8 * k) + 4 * lane;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (g + e >= Lin) d[e] = 0.f;
                }
            }
        }
    };

    // Two accumulator sets: tile t's transforms stay in set t & 1 while tile t + 1 is already being multiplied into the other set, so that
    // the exchange (LDS writes, a barrier, LDS reads, the output transform, the stores) rides under the next tile's MFMA stream.
    f32x16 acc[2][2];   // [tile parity][position sub-tile]
    // One half tile = 32 k-steps in 8 groups of 4.  The raw samples of group g + 1 are requested before group g's MFMAs are issued (two
    // register sets; the sched_barrier keeps the requests from sinking to their uses, where each MFMA pair would wait out an LDS round trip).
    auto mfma_half = [&](auto kind_tag, auto par_tag, auto half_tag, int buf, auto&& hook) {
        constexpr int KIND = decltype(kind_tag)::value, P = decltype(par_tag)::value, HALF = decltype(half_tag)::value;
        // Operand reads.  Row r of position sub-tile pm is pair 2 r + pm of the tile (not 32 pm + r): lane j's two pairs are neighbours and
        // need d0 .. d3 = window slots 4 j + 3 .. 4 j + 6 (pm 0) and 4 j + 5 .. 4 j + 8 (pm 1) of one input-channel row, so ONE aligned
        // 16-byte read q = slots 4 j + 4 .. 4 j + 7 (lanes at a 16-byte stride: conflict-free) carries
        //   transforms 1, 2 (KIND 1):  V = d2 +- d1 = q.y +- q.x (pm 0), q.w +- q.z (pm 1)                 -- nothing else to read
        //   transform 0     (KIND 0):  V = d0 - d2  = e.y - q.y  (pm 0), q.y - q.w (pm 1), e = the 8 bytes at slots 4 j + 2, 4 j + 3
        //   transform 3     (KIND 3):  V = d1 - d3  = q.x - q.z  (pm 0), q.z - e.x (pm 1), e = the 8 bytes at slots 4 j + 8, 4 j + 9
        // i.e. half an LDS instruction per MFMA for two of the four waves of a SIMD pair and one for the others.  (First version: two
        // ds_read_b32 per MFMA at an 8-byte lane stride, two-way bank conflicts; what that cost was not LDS bandwidth but LDS
        // INSTRUCTIONS -- two conflict-free ds_read_b64 per MFMA ran no faster, one per MFMA ran 6 % faster.)
        // (volatile: keeps every read its own instruction at an immediate offset; merged reads would each need a v_add for their base)
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef const volatile __attribute__((address_space(3))) f4* lds_vf4;
        typedef const volatile __attribute__((address_space(3))) f2* lds_vf2;
        const lds_vf4 xq = (lds_vf4)(s_x + buf * kHalfFloats + 4 * h * kXW + 4 * j + 4);
        const lds_vf2 xe = (lds_vf2)(s_x + buf * kHalfFloats + 4 * h * kXW + 4 * j + (KIND == 0 ? 2 : 8));
        // Three stages in flight: the reads of sub-step u + 2 are requested, the operands of sub-step u + 1 are formed (each in its own
        // register, between the MFMAs) while the MFMAs of sub-step u are issued.
        constexpr int SUB = LBM_WS_SUB, NU = 32 / SUB, RD = LBM_WS_RD;   // RD: sub-steps the reads run ahead of the MFMAs
        f4 tq[RD + 1][SUB];
        f2 te[RD + 1][SUB];
        float v[2][SUB][2];
        auto fetch = [&](int u, int set) {
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
                const int tt = u * SUB + c;
                const int off = (8 * (tt / 4) + (tt % 4)) * kXW;
                tq[set][c] = xq[off / 4];
                if (KIND != 1) te[set][c] = xe[off / 2];
            }
        };
        auto form = [&](int u, int c) {
            const f4 q = tq[u % (RD + 1)][c];
            float a, b;
            if (KIND == 1) { a = fmaf(sgn, q.x, q.y); b = fmaf(sgn, q.z, q.w); }
            else if (KIND == 0) { a = te[u % (RD + 1)][c].y - q.y; b = q.y - q.w; }
            else { a = q.x - q.z; b = q.z - te[u % (RD + 1)][c].x; }
            asm volatile("" : "+v"(a), "+v"(b));
            v[u & 1][c][0] = a; v[u & 1][c][1] = b;
        };
#pragma unroll
        for (int u = 0; u < RD; ++u) fetch(u, u);
#pragma unroll
        for (int c = 0; c < SUB; ++c) form(0, c);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (u + RD < NU) fetch(u + RD, (u + RD) % (RD + 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < SUB; ++c) {
                const int tt = u * SUB + c;
                if (HALF == 0 && tt == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][c][0], ureg[0], z, 0, 0, 0);
                    acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][c][1], ureg[0], z, 0, 0, 0);
                } else {
                    acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][c][0], ureg[HALF * 32 + tt], acc[P][0], 0, 0, 0);
                    acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u & 1][c][1], ureg[HALF * 32 + tt], acc[P][1], 0, 0, 0);
                }
                if (u + 1 < NU) form(u + 1, c);
                __builtin_amdgcn_sched_barrier(0);
            }
            // side work issued while these MFMAs occupy the pipe (once per four k-steps): window requests, the previous tile's exchange and outputs
            if (((u + 1) * SUB) % 4 == 0) hook(((u + 1) * SUB) / 4 - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // exchange area: [wave][pm][register quad rq][lane] float4 -- register quad rq of a lane = output pairs 8 rq + 4 h + (0..3) of column j.
    // Part q of 4 = the two register quads 2 (q & 1), 2 (q & 1) + 1 of position sub-tile q >> 1.
    auto ex_write = [&](auto par_tag, int q) {
        constexpr int P = decltype(par_tag)::value;
        const int pm = q >> 1;
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int rq = 2 * (q & 1) + r2;
            f4 v = {acc[P][pm][4 * rq], acc[P][pm][4 * rq + 1], acc[P][pm][4 * rq + 2], acc[P][pm][4 * rq + 3]};
            reinterpret_cast<f4*>(s_ex)[((wave * 2 + pm) * 4 + rq) * 64 + lane] = v;
        }
    };
    // wave (ti, c32) forms register quad ti of channel group c32.  Row r = 8 ti + 4 h + e holds pair 2 r + pm, so the two sub-tiles of a row
    // are four consecutive outputs and a lane's four rows 16 of them: four 16-byte stores, 64 contiguous bytes per lane.
    auto ex_output = [&](int64_t b, int pt) {
        // (row base: the wave-uniform part in 64 bits on the scalar unit, the lane's channel as one 24-bit multiply)
        float* yr = y + (b * Cout + (cg * 64 + c32 * 32)) * (int64_t)ldy + __umul24((unsigned)j, (unsigned)ldy);
        f4 m[2][4];
#pragma unroll
        for (int pm = 0; pm < 2; ++pm)
#pragma unroll
            for (int k = 0; k < 4; ++k) m[pm][k] = reinterpret_cast<const f4*>(s_ex)[(((k + 4 * c32) * 2 + pm) * 4 + ti) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f4 o;
            o.x = fmaxf(((m[0][0][e] + m[0][1][e]) + m[0][2][e]) + bv, lo);
            o.y = fmaxf(((m[0][1][e] - m[0][2][e]) - m[0][3][e]) + bv, lo);
            o.z = fmaxf(((m[1][0][e] + m[1][1][e]) + m[1][2][e]) + bv, lo);
            o.w = fmaxf(((m[1][1][e] - m[1][2][e]) - m[1][3][e]) + bv, lo);
            const int p = pt * 128 + 4 * (8 * ti + 4 * h + e);
            // ldy is a multiple of 4 >= Lin: the tail quad may spill into the row's pitch padding (as conv_wino3_kernel's does)
#if defined(LBM_WS_ABLATE) && (LBM_WS_ABLATE & 1)
            if (o.x + o.y + o.z + o.w == 123.456f) yr[p] = 0.f;   // (ablation: no stores)
#else
            if (p < Lin) *reinterpret_cast<f4*>(yr + p) = o;
#endif
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
#ifdef LBM_WS_STAMPS
    unsigned long long st_d[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_amdgcn_s_memtime();
#endif
    // one tile into accumulator set P; returns true after the last tile of the walk (whose exchange it then completes)
    auto tile_step = [&](auto kind_tag, auto par_tag) -> bool {
        constexpr int P = decltype(par_tag)::value;
        const std::integral_constant<int, 1 - P> other{};
        // half 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fix_edges(pt, 0);
        WS_STAMP(0);
        __syncthreads();           // half 0 of this tile is in buffer 0; every wave has read the outputs of the tile before the previous one
        WS_STAMP(1);
        // under half 0's MFMA stream: the requests for half 1 and the previous tile's transforms into the exchange area
        mfma_half(kind_tag, par_tag, std::integral_constant<int, 0>{}, 0, [&](int g) {
            if (g == 0) { request(b, pt, 1, 1, 0); request(b, pt, 1, 1, 1); }
            if (g == 1) { request(b, pt, 1, 1, 2); request(b, pt, 1, 1, 3); }
            if (g == 2) request(b, pt, 1, 1, 4);
            if (g < 4 && have_prev) ex_write(other, g);
        });
        WS_STAMP(4);
        // half 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fix_edges(pt, 1);
        WS_STAMP(5);
        __syncthreads();           // half 1 is in buffer 1; every wave is done with buffer 0; the previous tile's transforms are all in the exchange area
        WS_STAMP(6);
        int64_t nb = b + db;
        int npt = pt + dpt;
        if (npt >= ptiles) { npt -= ptiles; ++nb; }
        const int64_t ntile = tile + nstreams;
        const bool more = ntile < ntiles;
        // under half 1's MFMA stream: the requests for the next tile's half 0 and the previous tile's outputs
        mfma_half(kind_tag, par_tag, std::integral_constant<int, 1>{}, 1, [&](int g) {
            if (more) {
                if (g == 0) { request(nb, npt, 0, 0, 0); request(nb, npt, 0, 0, 1); }
                if (g == 1) { request(nb, npt, 0, 0, 2); request(nb, npt, 0, 0, 3); }
                if (g == 2) request(nb, npt, 0, 0, 4);
            }
#ifndef LBM_WS_OUT_HOOK
#define LBM_WS_OUT_HOOK 3
#endif
            if (g == LBM_WS_OUT_HOOK && have_prev) ex_output(pb, ppt);
        });
        WS_STAMP(8);
        pb = b; ppt = pt; have_prev = true;
        if (more) {
            tile = ntile; b = nb; pt = npt;
            return false;
        }
        __syncthreads();           // the exchange area is free again
#pragma unroll
        for (int q = 0; q < 4; ++q) ex_write(par_tag, q);
        __syncthreads();
        ex_output(pb, ppt);
        return true;
    };
    auto walk = [&](auto kind_tag) {
        while (true) {
            if (tile_step(kind_tag, std::integral_constant<int, 0>{})) break;
            if (tile_step(kind_tag, std::integral_constant<int, 1>{})) break;
        }
    };
#if defined(LBM_WS_ABLATE) && (LBM_WS_ABLATE & 8)
    walk(std::integral_constant<int, 1>{});   // (ablation: every wave runs the one-read walk)
    return;
#elif defined(LBM_WS_ABLATE) && (LBM_WS_ABLATE & 16)
    walk(std::integral_constant<int, 0>{});   // (ablation: every wave runs a two-read walk)
    return;
#endif
    if (ti == 0) walk(std::integral_constant<int, 0>{});          // wave-uniform: three instances of the walk that differ in the operand reads
    else if (ti == 3) walk(std::integral_constant<int, 3>{});
    else walk(std::integral_constant<int, 1>{});
#ifdef LBM_WS_STAMPS
    if (g_ws_stamps != nullptr && lane == 0) {
        unsigned long long* q = g_ws_stamps + ((size_t)blockIdx.x * 8 + wave) * 12;
        for (int i = 0; i < 12; ++i) q[i] = st_d[i];
    }
#endif
}

int g_cus[64];   // multiprocessor count per device (0 = not asked yet)

}  // namespace

#ifdef LBM_WS_STAMPS
extern "C" int lbm_debug_ws_stamps(unsigned long long* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), &p, sizeof(p)); }
#endif

namespace lbm {

bool wino3_ws_eligible(const float* x, int64_t ldx, const float* y, int64_t ldy, int Cin, int Cout) {
    return Cin == kCin && Cout % 64 == 0 && Cout >= 64 && ldx % 4 == 0 && ldy % 4 == 0 && ldy < (1 << 24) &&
           ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
}

int launch_wino3_ws(const float* x, const float* utab, const float* bias, float* y, int64_t B, int Lin, int64_t ldx, int Cout, int64_t ldy,
                    int relu, hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail(LBM_ELAUNCH, "lbm_conv1d_wino3_f32: hipGetDevice failed");
    if (g_cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino3_ws_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kLdsBytes);
        if (e != hipSuccess) return fail(LBM_ELAUNCH, "lbm_conv1d_wino3_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        g_cus[dev] = n;
    }
    const int ptiles = (Lin + 127) / 128;
    const int64_t ntiles = B * ptiles;
    const int ncg = Cout / 64, mtiles = (Cout + 127) / 128;
    // one workgroup per CU (137 KB of LDS each): streams = CUs / channel groups, a multiple of 8 where possible (XCD-aware dealing)
    int nstreams = g_cus[dev] / ncg;
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
    hipLaunchKernelGGL(conv_wino3_ws_kernel, dim3((unsigned)(nstreams * ncg)), dim3(kThreads), kLdsBytes, st, x, utab, bias, y, (int)ldx, (int)ldy,
                       Lin, Cout, relu, ptiles, ntiles, mtiles, ncg, nstreams, xcd_map);
    return check_launch("lbm_conv1d_wino3_f32");
}

}  // namespace lbm
