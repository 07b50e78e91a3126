// Conv1d (+bias, +ReLU) of the VQ-VAE encoder (a10) as an implicit GEMM on fp32 MFMA.
//
// y[b, co, p] = bias[co] + sum_{ci,kt} w[co, ci, kt] * x[b, ci, p*stride - pad + kt]
// GEMM view per epoch: positions x output channels, K = Cin*ksize; the weights viewed as [Cout][Cin*ksize]
// (torch's layout already is that row-major matrix).  im2col is never materialised.  Rows of x and y may
// be pitched (ldx, ldy floats per row): with pitches that are multiples of 4 every global access is a
// 16-byte one, which is what the pipeline's internal activation buffers use.
//
// conv_tile_kernel<KS, STRIDE, BM, VEC> (the encoder's four layer shapes):
//   workgroup tile = 128 positions of ONE epoch x BM output channels, 4 wavefronts as 2x2, wave tile 64 x BM/2,
//   v_mfma_f32_32x32x2_f32 with the POSITIONS as the M operand and the channels as N, so an accumulator
//   register quad is 4 consecutive positions of one channel and the epilogue stores 16 bytes per lane.
//   K is consumed CK input channels at a time.  Per chunk the LDS holds the weight slab [BM][CK*KS(+4)] and
//   the raw input rows [CK][window] (window = 16-byte aligned span of the 128 positions' receptive field):
//   the operand for tap kt of channel ci is read straight from the row at p*STRIDE+kt, so the KS-fold
//   im2col replication never exists and global loads are contiguous row segments.
//   Software pipeline with ONE barrier per chunk: while chunk ch is multiplied out of LDS image ch&1,
//   chunk ch+1 (in registers since the previous iteration) is written to the other image and chunk ch+2's
//   global loads are issued, both in the middle of the MFMA stream.
//   VEC = 4: 16-byte global loads/stores (aligned pitches); VEC = 1: same structure, 4-byte accesses.
// conv1d_generic_kernel: any ksize/stride/pad (runtime), simple single-buffered gather; used for shapes
//   outside the instantiations above.
// fp32 MFMA = exact fp32 fmaf chain; summation order: chunks ascending, inside a chunk the two lane
// halves own the first/second half of the chunk's (ci, kt) list in ascending order.
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "conv_ws.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BN = 128;

// PM = 32-position MFMA sub-tiles per wave: the workgroup tile is 64 PM positions (2 waves along the positions).  PM = 2 is the
// 128-position tile every layer runs with.  Round 3 measured PM = 3 / 4 on the stride-2 layers (192 / 256 positions per tile, to spread the
// fixed per-tile cost the stamps show: profiles/r03_experiments.md): conv1 2.39 -> 2.35 ms, conv2 3.61 -> 3.59 ms, i.e. within noise; not used.
template <int KS, int STRIDE, int BM, int PM = 2>
struct ConvCfg {
    static constexpr int BN = 64 * PM;                   // positions per workgroup tile
    static constexpr int CK = (KS == 1) ? 16 : 8;        // input channels per chunk
    static constexpr int KC = CK * KS;                   // GEMM-K per chunk (16 / 24 / 32)
    static constexpr int LDA = KC + 4;                   // padded weight-slab row (conflict-free ds_read_b128)
    static constexpr int PADMAX = (KS == 1) ? 0 : 4;     // the window starts 4 samples early when pad > 0 (pad <= 4)
    static constexpr int XW = ((BN - 1) * STRIDE + KS + PADMAX + 3) & ~3;   // window samples per channel row (BN: this struct's)
    static constexpr int X4 = XW / 4;
    static constexpr int A_F4 = BM * KC / 4 / kThreads;  // weight float4 per thread per chunk
    static constexpr int X_F4 = (CK * X4 + kThreads - 1) / kThreads;   // input quads per thread per chunk
    static constexpr int CN = BM / 64;                   // 32-channel MFMA tiles per wave (wave tile = 64 pos x BM/2)
    static constexpr size_t LDS_FLOATS = 2 * (size_t)(BM * LDA + CK * XW);
    static_assert(KC % 8 == 0, "each lane half consumes KC/2 k-steps in groups of 4");
    static_assert((BM * KC / 4) % kThreads == 0, "weight slab must split evenly");
};

template <int VEC>
__device__ __forceinline__ float4 load_quad(const float* p, int o0, int o1, int o2, int o3) {
    if constexpr (VEC == 4) return *reinterpret_cast<const float4*>(p + o0);
    else return make_float4(p[o0], p[o1], p[o2], p[o3]);
}

template <int KS, int STRIDE, int BM, int VEC, int PM = 2>
__global__ __launch_bounds__(kThreads) void conv_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int Cin, int Lin, int ldx, int Cout, int Lout, int ldy,
                                                             int pad, int relu, int ptiles, int mtiles) {
    using Cfg = ConvCfg<KS, STRIDE, BM, PM>;
    constexpr int CK = Cfg::CK, KC = Cfg::KC, LDA = Cfg::LDA, XW = Cfg::XW, X4 = Cfg::X4, CN = Cfg::CN, BN = Cfg::BN;
    extern __shared__ __align__(16) float smem[];
    float* const s_a0 = smem;                 // 2 x [BM][LDA]   weights
    float* const s_x0 = smem + 2 * BM * LDA;  // 2 x [CK][XW]    input windows

    const int Ktot = Cin * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> (channel tile fastest, then position tile, then epoch): blocks sharing an input tile are neighbours
    const int mt = blockIdx.x % mtiles;
    const int rest = blockIdx.x / mtiles;
    const int pt = rest % ptiles;
    const int64_t b = rest / ptiles;
    const int m0 = mt * BM, p0 = pt * BN;
    const int pada = pad > 0 ? Cfg::PADMAX : 0;
    const int ga = p0 * STRIDE - pada;        // first sample of the window (multiple of 4)
    const float* xb = x + b * (int64_t)Cin * ldx;

    float bv[CN];
#pragma unroll
    for (int cn = 0; cn < CN; ++cn) {
        const int co = m0 + wm * (BM / 2) + cn * 32 + j;
        bv[cn] = 0.f;
        if (bias != nullptr && co < Cout) bv[cn] = bias[co];   // first used in the epilogue
    }

    float4 ra[Cfg::A_F4];
    float4 rx[Cfg::X_F4];

    // Loop-invariant staging coordinates of this thread (hoisted: the chunk loop only adds a uniform base).
    int a_off[Cfg::A_F4], a_lds[Cfg::A_F4], a_k[Cfg::A_F4];
    bool a_ok[Cfg::A_F4];
#pragma unroll
    for (int i = 0; i < Cfg::A_F4; ++i) {
        const int f = threadIdx.x + i * kThreads;  // float4 index in the [BM][KC/4] slab
        const int m = f / (KC / 4), c4 = f - m * (KC / 4);
        a_ok[i] = m0 + m < Cout;
        a_off[i] = (a_ok[i] ? m0 + m : 0) * Ktot + 4 * c4;
        a_lds[i] = m * LDA + 4 * c4;
        a_k[i] = 4 * c4;
    }
    int x_o[Cfg::X_F4][VEC == 4 ? 1 : 4], x_lds[Cfg::X_F4];
    bool x_ok[Cfg::X_F4][4];
#pragma unroll
    for (int i = 0; i < Cfg::X_F4; ++i) {
        const int f = threadIdx.x + i * kThreads;
        const bool slot = f < CK * X4;
        const int r = slot ? f / X4 : 0, c4 = slot ? f - r * X4 : 0;
        const int g = ga + 4 * c4;
        x_lds[i] = r * XW + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) x_ok[i][e] = slot && (unsigned)(g + e) < (unsigned)Lin;
        if constexpr (VEC == 4) {
            // g is a multiple of 4 and so is ldx: a clamped quad is either fully outside [0, Lin) or inside the pitch
            x_o[i][0] = r * ldx + (g < 0 ? 0 : (g > ldx - 4 ? ldx - 4 : g));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x_o[i][e] = r * ldx + (g + e < 0 ? 0 : (g + e < Lin ? g + e : Lin - 1));
        }
    }

    // Chunk ch covers input channels [ch*CK, ch*CK+CK).  A ragged last chunk is shifted down to end at Cin
    // (needs Cin >= CK) and the weights of the channels it shares with the previous chunk are zeroed, so
    // every x row read is a real row.  Loads are unconditional at clamped addresses; the masks are applied
    // when the registers are written to LDS (a "load or zero" select makes the compiler branch around every
    // load and drain vmcnt per element).
    auto load_chunk = [&](int ch) {
        const int cn = ch * CK;
        const int ci0 = cn + CK <= Cin ? cn : Cin - CK;
        const float* wc = w + ci0 * KS;
        const float* xc = xb + (int64_t)ci0 * ldx;
#pragma unroll
        for (int i = 0; i < Cfg::A_F4; ++i) ra[i] = *reinterpret_cast<const float4*>(wc + a_off[i]);
#pragma unroll
        for (int i = 0; i < Cfg::X_F4; ++i) {
            if constexpr (VEC == 4) rx[i] = load_quad<4>(xc, x_o[i][0], 0, 0, 0);
            else rx[i] = load_quad<1>(xc, x_o[i][0], x_o[i][1], x_o[i][2], x_o[i][3]);
        }
    };
    // Writes the registers loaded by load_chunk(ch) to LDS image `buf`; the empty asm pins each loaded value
    // so the compiler cannot sink a load back under its mask.
    auto store_chunk = [&](int buf, int ch) {
        const int cn = ch * CK;
        const int shift = (cn + CK <= Cin ? 0 : cn + CK - Cin) * KS;   // k entries the shifted chunk re-reads
        float* sa = s_a0 + buf * (BM * LDA);
        float* sx = s_x0 + buf * (CK * XW);
#pragma unroll
        for (int i = 0; i < Cfg::A_F4; ++i) {
            asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
            // (rows past Cout hold row 0's weights: their outputs are never stored, so they need no mask)
            if (shift == 0) *reinterpret_cast<float4*>(sa + a_lds[i]) = ra[i];   // uniform
            else *reinterpret_cast<float4*>(sa + a_lds[i]) = a_k[i] >= shift ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_F4; ++i) {
            asm volatile("" : "+v"(rx[i].x), "+v"(rx[i].y), "+v"(rx[i].z), "+v"(rx[i].w));
            if (threadIdx.x + i * kThreads < CK * X4)
                *reinterpret_cast<float4*>(sx + x_lds[i]) =
                    make_float4(x_ok[i][0] ? rx[i].x : 0.f, x_ok[i][1] ? rx[i].y : 0.f, x_ok[i][2] ? rx[i].z : 0.f,
                                x_ok[i][3] ? rx[i].w : 0.f);
        }
    };

    f32x16 acc[PM][CN];   // [position sub-tile][channel sub-tile]
#pragma unroll
    for (int pm = 0; pm < PM; ++pm)
#pragma unroll
        for (int cn = 0; cn < CN; ++cn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pm][cn][r] = 0.f;

    // k-steps [4*qlo, 4*qhi) of this lane half's list, from LDS image `buf`
    const int xsh = (wn * (32 * PM) + j) * STRIDE + (pada - pad);
    auto mfma_block = [&](int buf, int qlo, int qhi) {
        const float* wp = s_a0 + buf * (BM * LDA) + (wm * (BM / 2) + j) * LDA + h * (KC / 2);
        const float* xp = s_x0 + buf * (CK * XW) + (h * (CK / 2)) * XW + xsh;
#pragma unroll
        for (int q = qlo; q < qhi; ++q) {
            float wv[CN][4];
#pragma unroll
            for (int cn = 0; cn < CN; ++cn) {
                const float4 t = *reinterpret_cast<const float4*>(wp + cn * 32 * LDA + 4 * q);
                wv[cn][0] = t.x; wv[cn][1] = t.y; wv[cn][2] = t.z; wv[cn][3] = t.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;            // k-step inside this lane half's list
                const int ci = s / KS, kt = s % KS;  // compile-time after unrolling
                float xs[PM];
#pragma unroll
                for (int pm = 0; pm < PM; ++pm) xs[pm] = xp[ci * XW + kt + pm * 32 * STRIDE];
#pragma unroll
                for (int cn = 0; cn < CN; ++cn)
#pragma unroll
                    for (int pm = 0; pm < PM; ++pm)
                        acc[pm][cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[pm], wv[cn][u], acc[pm][cn], 0, 0, 0);
            }
        }
    };

    constexpr int QN = KC / 8, QH = QN / 2;
    const int nchunks = (Cin + CK - 1) / CK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    load_chunk(nchunks > 1 ? 1 : 0);
    // The first chunk is peeled out of the loop: its first products then START the accumulators (MFMA with a literal zero C)
    // instead of adding to 128 registers that had to be cleared one by one.
    auto chunk_step = [&](int ch) {
        const int buf = ch & 1;
        mfma_block(buf, 0, QH);
        store_chunk(buf ^ 1, ch + 1);
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);   // the tail re-reads the last chunk (unused)
        mfma_block(buf, QH, QN);
        __syncthreads();
    };
    if (nchunks > 1) {
        chunk_step(0);
        for (int ch = 1; ch + 1 < nchunks; ++ch) chunk_step(ch);
    }
    mfma_block((nchunks - 1) & 1, 0, QN);

    // Epilogue: accumulator register quad rq of a lane = positions 8*rq + 4*h + (0..3) of channel column j.
    float* yb = y + b * (int64_t)Cout * ldy;
    const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int cn = 0; cn < CN; ++cn) {
        const int co = m0 + wm * (BM / 2) + cn * 32 + j;
        if (co >= Cout) continue;
        float* yr = yb + (int64_t)co * ldy;
#pragma unroll
        for (int pm = 0; pm < PM; ++pm) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int p = p0 + wn * (32 * PM) + pm * 32 + 8 * rq + 4 * h;
                const float v0 = fmaxf(acc[pm][cn][4 * rq + 0] + bv[cn], lo), v1 = fmaxf(acc[pm][cn][4 * rq + 1] + bv[cn], lo);
                const float v2 = fmaxf(acc[pm][cn][4 * rq + 2] + bv[cn], lo), v3 = fmaxf(acc[pm][cn][4 * rq + 3] + bv[cn], lo);
                if constexpr (VEC == 4) {
                    // p + 3 < ldy because ldy is a multiple of 4 >= Lout: the tail quad may spill into the row's pitch padding
                    if (p < Lout) *reinterpret_cast<float4*>(yr + p) = make_float4(v0, v1, v2, v3);
                } else {
                    if (p < Lout) yr[p] = v0;
                    if (p + 1 < Lout) yr[p + 1] = v1;
                    if (p + 2 < Lout) yr[p + 2] = v2;
                    if (p + 3 < Lout) yr[p + 3] = v3;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------- Winograd F(2,3) path
// conv_wino3_kernel<VEC>: ksize 3, stride 1, pad 1 (the encoder's 128 -> 256 layer) with the minimal-filtering
// identity F(2,3): for the output pair (2j, 2j+1) and d_k = x[2j-1+k],
//   V0 = d0-d2, V1 = d1+d2, V2 = d2-d1, V3 = d1-d3;  U0 = g0, U1 = (g0+g1+g2)/2, U2 = (g0-g1+g2)/2, U3 = g2
//   M_i = sum_ci U_i[co,ci] * V_i[ci,j];   y[2j] = M0+M1+M2,  y[2j+1] = M1-M2-M3
// i.e. four GEMMs with K = Cin over HALF the positions: 2/3 of the direct form's MFMA work.  Same tile and
// pipeline as conv_tile_kernel (128 positions = 64 pairs x 128 channels, pairs as the MFMA M operand); the
// V's are formed on the fly from four raw-window reads per input channel, the U's when the weight slab is
// staged.  fp32 throughout; differs from the direct form by a few ulp of |x||w| (the same order as the
// difference between two BLAS libraries; vendor conv libraries use this transform for 3-tap fp32 convs).
// Pre-transformed weights of the Winograd kernel (lbm_conv1d_wino3_tables_init): U in the order the staging threads consume it,
// [chunk of 8 input channels][tile of 128 output channels][transform i][thread tid = 2 co + quad][4 floats] -- one chunk's slab is four
// fully coalesced 16-byte loads per thread instead of three 48-byte-strided ones plus the transform (stamps, round 2: issuing
// those loads cost a wave ~990 of a chunk's ~5 700 cycles; the table's cost ~250).  Rows past Cout are zero.
__global__ __launch_bounds__(256) void wino3_table_kernel(const float* __restrict__ w, float* __restrict__ tab, int Cin, int Cout, int mtiles) {
    const int tid = threadIdx.x, ch = blockIdx.x, mt = blockIdx.y;
    const int co = mt * 128 + (tid >> 1), cq = tid & 1;
    float u[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float* g = w + ((int64_t)(co < Cout ? co : 0) * Cin + ch * 8 + 4 * cq + c) * 3;
        const float g0 = co < Cout ? g[0] : 0.f, g1 = co < Cout ? g[1] : 0.f, g2 = co < Cout ? g[2] : 0.f;
        u[0][c] = g0;
        u[1][c] = ((g0 + g2) + g1) * 0.5f;
        u[2][c] = ((g0 + g2) - g1) * 0.5f;
        u[3][c] = g2;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
        reinterpret_cast<float4*>(tab)[(((int64_t)ch * mtiles + mt) * 4 + i) * 256 + tid] = make_float4(u[i][0], u[i][1], u[i][2], u[i][3]);
}

struct WinoCfg {
    static constexpr int BM = 128, CK = 8;
    static constexpr int LDU = CK + 4;                   // padded U row (conflict-free ds_read_b128)
    static constexpr int XW = ((BN - 1) + 3 + 4 + 3) & ~3;   // window samples per channel row (as ConvCfg<3,1>)
    static constexpr int X4 = XW / 4;
    static constexpr int X_F4 = (CK * X4 + kThreads - 1) / kThreads;
    static constexpr int USZ = 4 * BM * LDU, XSZ = CK * XW;
    static constexpr size_t LDS_FLOATS = 2 * (size_t)(USZ + XSZ);
};

// UTAB: w is the pre-transformed table above instead of the [Cout, Cin, 3] weights (same U values, so the same results bit for bit)
template <int VEC, bool UTAB>
__global__ __launch_bounds__(kThreads) void conv_wino3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int Cin, int Lin, int ldx, int Cout, int Lout, int ldy,
                                                              int relu, int ptiles, int mtiles) {
    using Cfg = WinoCfg;
    constexpr int BM = Cfg::BM, CK = Cfg::CK, LDU = Cfg::LDU, XW = Cfg::XW, X4 = Cfg::X4, CN = 2;
    extern __shared__ __align__(16) float smem[];
    float* const s_u0 = smem;                  // 2 x [4][BM][LDU]  transformed weights
    float* const s_x0 = smem + 2 * Cfg::USZ;   // 2 x [CK][XW]      raw input windows

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int mt = blockIdx.x % mtiles;
    const int rest = blockIdx.x / mtiles;
    const int pt = rest % ptiles;
    const int64_t b = rest / ptiles;
    const int m0 = mt * BM, p0 = pt * BN;
    const int ga = p0 - 4;                     // first sample of the window (pad = 1, aligned start 4 early)
    const float* xb = x + b * (int64_t)Cin * ldx;

    float bv[CN];
#pragma unroll
    for (int cn = 0; cn < CN; ++cn) {
        const int co = m0 + wm * 64 + cn * 32 + j;
        bv[cn] = 0.f;
        if (bias != nullptr && co < Cout) bv[cn] = bias[co];
    }

    // weights: thread -> (channel co = tid / 2, input-channel quad cq = tid & 1): 12 consecutive floats
    // w[co][ci0 + 4 cq .. +3][0..2] = three 16-byte loads (Cin % 4 == 0 keeps them aligned)
    const int u_co = threadIdx.x >> 1, u_cq = threadIdx.x & 1;
    const bool u_ok = m0 + u_co < Cout;
    const int u_off = ((u_ok ? m0 + u_co : 0) * Cin + 4 * u_cq) * 3;
    float4 rw[UTAB ? 4 : 3];
    float4 rx[Cfg::X_F4];
    int x_o[Cfg::X_F4][VEC == 4 ? 1 : 4], x_lds[Cfg::X_F4];
    bool x_ok[Cfg::X_F4][4];
#pragma unroll
    for (int i = 0; i < Cfg::X_F4; ++i) {
        const int f = threadIdx.x + i * kThreads;
        const bool slot = f < CK * X4;
        const int r = slot ? f / X4 : 0, c4 = slot ? f - r * X4 : 0;
        const int g = ga + 4 * c4;
        x_lds[i] = r * XW + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) x_ok[i][e] = slot && (unsigned)(g + e) < (unsigned)Lin;
        if constexpr (VEC == 4) {
            x_o[i][0] = r * ldx + (g < 0 ? 0 : (g > ldx - 4 ? ldx - 4 : g));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) x_o[i][e] = r * ldx + (g + e < 0 ? 0 : (g + e < Lin ? g + e : Lin - 1));
        }
    }

    auto load_chunk = [&](int ch) {
        const float* xc = xb + (int64_t)ch * CK * ldx;
        if constexpr (UTAB) {
            const float4* ut = reinterpret_cast<const float4*>(w) + ((int64_t)(ch * mtiles + mt) * 4) * kThreads + threadIdx.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) rw[i] = ut[i * kThreads];
        } else {
            const float* wc = w + ch * CK * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) rw[i] = *reinterpret_cast<const float4*>(wc + u_off + 4 * i);
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_F4; ++i) {
            if constexpr (VEC == 4) rx[i] = load_quad<4>(xc, x_o[i][0], 0, 0, 0);
            else rx[i] = load_quad<1>(xc, x_o[i][0], x_o[i][1], x_o[i][2], x_o[i][3]);
        }
    };
    auto store_chunk = [&](int buf) {
        float* su = s_u0 + buf * Cfg::USZ + u_co * LDU + 4 * u_cq;
        float* sx = s_x0 + buf * Cfg::XSZ;
        if constexpr (UTAB) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("" : "+v"(rw[i].x), "+v"(rw[i].y), "+v"(rw[i].z), "+v"(rw[i].w));
                *reinterpret_cast<float4*>(su + i * (BM * LDU)) = rw[i];
            }
        } else {
    #pragma unroll
            for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(rw[i].x), "+v"(rw[i].y), "+v"(rw[i].z), "+v"(rw[i].w));
            const float g[12] = {rw[0].x, rw[0].y, rw[0].z, rw[0].w, rw[1].x, rw[1].y, rw[1].z, rw[1].w,
                                 rw[2].x, rw[2].y, rw[2].z, rw[2].w};   // [ci 0..3][tap 0..2]
            float u[4][4];
    #pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float g0 = g[3 * c], g1 = g[3 * c + 1], g2 = g[3 * c + 2];   // (rows past Cout: row 0's weights, never stored)
                u[0][c] = g0;
                u[1][c] = ((g0 + g2) + g1) * 0.5f;
                u[2][c] = ((g0 + g2) - g1) * 0.5f;
                u[3][c] = g2;
            }
    #pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(su + i * (BM * LDU)) = make_float4(u[i][0], u[i][1], u[i][2], u[i][3]);
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_F4; ++i) {
            asm volatile("" : "+v"(rx[i].x), "+v"(rx[i].y), "+v"(rx[i].z), "+v"(rx[i].w));
            if (threadIdx.x + i * kThreads < CK * X4)
                *reinterpret_cast<float4*>(sx + x_lds[i]) =
                    make_float4(x_ok[i][0] ? rx[i].x : 0.f, x_ok[i][1] ? rx[i].y : 0.f, x_ok[i][2] ? rx[i].z : 0.f,
                                x_ok[i][3] ? rx[i].w : 0.f);
        }
    };

    f32x16 acc[4][CN];   // [transform i][channel sub-tile]; rows = output pairs
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cn = 0; cn < CN; ++cn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][cn][r] = 0.f;

    // input channels h*4 + s, s in [slo, shi), of LDS image `buf`.  Pair jl = wn*32 + j of the tile needs the
    // samples 2 jl - 1 .. 2 jl + 2 relative to p0, i.e. window slots 2 jl + 3 .. 2 jl + 6.
    const int xsh = 2 * (wn * 32 + j) + 3;
    auto mfma_block = [&](int buf, int slo, int shi) {
        const float* up = s_u0 + buf * Cfg::USZ + (wm * 64 + j) * LDU + h * 4;
        const float* xp = s_x0 + buf * Cfg::XSZ + (h * 4) * XW + xsh;
        float uv[4][CN][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cn = 0; cn < CN; ++cn) {
                const float4 t = *reinterpret_cast<const float4*>(up + i * (BM * LDU) + cn * 32 * LDU);
                uv[i][cn][0] = t.x; uv[i][cn][1] = t.y; uv[i][cn][2] = t.z; uv[i][cn][3] = t.w;
            }
#pragma unroll
        for (int s = slo; s < shi; ++s) {
            const float d0 = xp[s * XW], d1 = xp[s * XW + 1], d2 = xp[s * XW + 2], d3 = xp[s * XW + 3];
            // (each V in its own register, pinned: formed one by one in front of their MFMAs they went through ONE register and every
            // v_sub waited for the previous MFMA to have read it -- tools/asm_war_scan.py, round 3: 36 of a chunk's 64 MFMAs)
            float v[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cn = 0; cn < CN; ++cn)
                    acc[i][cn] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[i], uv[i][cn][s], acc[i][cn], 0, 0, 0);
        }
    };

    const int nchunks = Cin / CK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    load_chunk(nchunks > 1 ? 1 : 0);
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        const int buf = ch & 1;
        mfma_block(buf, 0, 2);
        __builtin_amdgcn_sched_barrier(0);   // keep the wait for chunk ch+1's loads behind the first half's MFMAs
        store_chunk(buf ^ 1);
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);
        mfma_block(buf, 2, 4);
        __syncthreads();
    }
    mfma_block((nchunks - 1) & 1, 0, 4);

    // Epilogue: accumulator register quad rq of a lane = output pairs 8*rq + 4*h + (0..3) of channel column j,
    // i.e. 8 consecutive positions: two 16-byte stores.
    float* yb = y + b * (int64_t)Cout * ldy;
    const float lo = relu ? 0.f : -INFINITY;
#pragma unroll
    for (int cn = 0; cn < CN; ++cn) {
        const int co = m0 + wm * 64 + cn * 32 + j;
        if (co >= Cout) continue;
        float* yr = yb + (int64_t)co * ldy;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * rq + e;
                const float m0v = acc[0][cn][r], m1 = acc[1][cn][r], m2 = acc[2][cn][r], m3 = acc[3][cn][r];
                o[2 * e] = fmaxf(((m0v + m1) + m2) + bv[cn], lo);
                o[2 * e + 1] = fmaxf(((m1 - m2) - m3) + bv[cn], lo);
            }
            const int p = p0 + 2 * (wn * 32 + 8 * rq + 4 * h);
            if constexpr (VEC == 4) {
                if (p < Lout) *reinterpret_cast<float4*>(yr + p) = make_float4(o[0], o[1], o[2], o[3]);
                if (p + 4 < Lout) *reinterpret_cast<float4*>(yr + p + 4) = make_float4(o[4], o[5], o[6], o[7]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (p + e < Lout) yr[p + e] = o[e];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- generic path
constexpr int GBM = 64, GKC = 32;
constexpr int GLDA = GKC + 4;

__global__ __launch_bounds__(kThreads) void conv1d_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int64_t Ntot, int Cin, int Lin, int64_t ldx, int Cout, int Lout,
                                                                  int64_t ldy, int ks, int stride, int pad, int relu) {
    __shared__ __align__(16) float s_a[GBM * GLDA];
    __shared__ __align__(16) float s_b[GKC * BN];

    const int Ktot = Cin * ks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GBM;
    const int64_t n0 = (int64_t)blockIdx.x * BN;

    const int nb = threadIdx.x & (BN - 1);
    const int kb = threadIdx.x >> 7;
    const int64_t ncol = n0 + nb;
    const bool col_ok = ncol < Ntot;
    const int64_t bcol = col_ok ? ncol / Lout : 0;
    const int pcol = col_ok ? (int)(ncol - bcol * Lout) : 0;
    const float* xcol = x + bcol * (int64_t)Cin * ldx;
    const int pos0 = pcol * stride - pad;
    const int ak = threadIdx.x & 31, am = threadIdx.x >> 5;

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = 0.f;
        acc1[r] = 0.f;
    }
    int ci_run = kb / ks, kt_run = kb - ci_run * ks;

    for (int kk0 = 0; kk0 < Ktot; kk0 += GKC) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < GBM / 8; ++i) {
            const int m = am + 8 * i;
            const int co = m0 + m, kk = kk0 + ak;
            s_a[m * GLDA + ak] = (co < Cout && kk < Ktot) ? w[(int64_t)co * Ktot + kk] : 0.f;
        }
        {
            int ci = ci_run, kt = kt_run;
#pragma unroll
            for (int i = 0; i < GKC / 2; ++i) {
                const int kl = kb + 2 * i;
                float v = 0.f;
                const int pos = pos0 + kt;
                if (col_ok && ci < Cin && pos >= 0 && pos < Lin) v = xcol[(int64_t)ci * ldx + pos];
                s_b[kl * BN + nb] = v;
                kt += 2;
                while (kt >= ks) {
                    kt -= ks;
                    ++ci;
                }
            }
            ci_run = ci;
            kt_run = kt;
        }
        __syncthreads();

        const float* ap = s_a + (wm * 32 + j) * GLDA + h * (GKC / 2);
        const float* bp = s_b + (h * (GKC / 2)) * BN + wn * 64 + j;
#pragma unroll
        for (int q = 0; q < GKC / 8; ++q) {
            const float4 a = reinterpret_cast<const float4*>(ap)[q];
            const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bp[s * BN], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bp[s * BN + 32], acc1, 0, 0, 0);
            }
        }
    }

#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        const int64_t n = n0 + wn * 64 + sub * 32 + j;
        if (n >= Ntot) continue;
        const int64_t b = n / Lout;
        const int p = (int)(n - b * Lout);
        float* yb = y + b * (int64_t)Cout * ldy + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < Cout) {
                float v = (sub == 0 ? acc0[r] : acc1[r]) + (bias != nullptr ? bias[co] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                yb[(int64_t)co * ldy] = v;
            }
        }
    }
}

template <int KS, int STRIDE, int BM, int PM = 2>
int launch_tile(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout,
                int Lout, int64_t ldy, int pad, int relu, hipStream_t st) {
    using Cfg = ConvCfg<KS, STRIDE, BM, PM>;
    const int ptiles = (Lout + Cfg::BN - 1) / Cfg::BN;
    const int mtiles = (Cout + BM - 1) / BM;
    const int64_t blocks = B * ptiles * mtiles;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    const size_t lds = Cfg::LDS_FLOATS * sizeof(float);
    const bool vec = (ldx % 4 == 0) && (ldy % 4 == 0) && ldx >= 4 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    auto k = vec ? conv_tile_kernel<KS, STRIDE, BM, 4, PM> : conv_tile_kernel<KS, STRIDE, BM, 1, PM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, w, bias, y, Cin, Lin, (int)ldx, Cout, Lout, (int)ldy,
                       pad, relu, ptiles, mtiles);
    return lbm::check_launch("lbm_conv1d_f32");
}

int launch_wino3(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout,
                 int Lout, int64_t ldy, int relu, hipStream_t st, bool utab = false) {
    const int ptiles = (Lout + BN - 1) / BN;
    const int mtiles = (Cout + WinoCfg::BM - 1) / WinoCfg::BM;
    const int64_t blocks = B * ptiles * mtiles;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    const size_t lds = WinoCfg::LDS_FLOATS * sizeof(float);
    const bool vec = (ldx % 4 == 0) && (ldy % 4 == 0) && ldx >= 4 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0;
    auto k = utab ? (vec ? conv_wino3_kernel<4, true> : conv_wino3_kernel<1, true>) : (vec ? conv_wino3_kernel<4, false> : conv_wino3_kernel<1, false>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, w, bias, y, Cin, Lin, (int)ldx, Cout, Lout, (int)ldy,
                       relu, ptiles, mtiles);
    return lbm::check_launch("lbm_conv1d_f32");
}

}  // namespace

extern "C" int lbm_conv1d_pitched_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy,
                                      int64_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride,
                                      int32_t pad, int32_t relu, lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Lin >= 1, LBM_EINVAL, "lbm_conv1d_f32: bad sizes");
    LBM_REQUIRE(ksize >= 1 && stride >= 1 && pad >= 0, LBM_EINVAL, "lbm_conv1d_f32: bad ksize/stride/pad");
    LBM_REQUIRE(Lin + 2 * pad >= ksize, LBM_EINVAL, "lbm_conv1d_f32: kernel (%d) larger than padded input (%d)", ksize, Lin + 2 * pad);
    const int Lout = (Lin + 2 * pad - ksize) / stride + 1;
    LBM_REQUIRE(ldx >= Lin && ldy >= Lout, LBM_EINVAL, "lbm_conv1d_f32: row pitch smaller than the row (ldx=%lld Lin=%d ldy=%lld Lout=%d)",
                (long long)ldx, Lin, (long long)ldy, Lout);
    LBM_REQUIRE((int64_t)Cin * ldx <= 0x7fffffffLL && (int64_t)Cout * ldy <= 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_conv1d_f32: one epoch exceeds 2^31 elements");
    if (B == 0) return LBM_OK;
    LBM_REQUIRE(x && w && y, LBM_EINVAL, "lbm_conv1d_f32: NULL pointer");
    hipStream_t st = lbm::as_stream(stream);
    const bool w_vec = (((uintptr_t)w & 15) == 0) && ((Cin * ksize) % 4 == 0);
    const int ck = ksize == 1 ? 16 : 8;   // ConvCfg::CK: a ragged last chunk is shifted, which needs Cin >= CK
    // the encoder's two 4-tap stride-2 layers with 16-byte rows: weights-stationary form (conv_k4s2_ws.hip), bit-identical results.
    // LBM_K4S2_KERNEL=tile (read once) keeps the tile kernel for A/B measurements.
    static const bool k4_ws_off = [] { const char* v = getenv("LBM_K4S2_KERNEL"); return v != nullptr && strcmp(v, "tile") == 0; }();
    if (!k4_ws_off && ksize == 4 && stride == 2 && pad == 1 && lbm::k4s2_ws_eligible(x, ldx, w, y, ldy, Cin, Cout, Lin))
        return lbm::launch_k4s2_ws(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, relu, st);
    if (w_vec && Cin >= ck && pad <= (ksize == 1 ? 0 : 4)) {
        if (ksize == 4 && stride == 2 && Cout <= 64) return launch_tile<4, 2, 64>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
        if (ksize == 4 && stride == 2) return launch_tile<4, 2, 128>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
        if (ksize == 3 && stride == 1 && pad == 1 && Cin % 8 == 0 && Cout >= 64)   // Winograd F(2,3): 2/3 of the MFMA work
            return launch_wino3(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, relu, st);
        if (ksize == 3 && stride == 1 && Cin % 4 == 0 && Cout >= 256) return launch_tile<3, 1, 256>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
        if (ksize == 1 && stride == 1 && Cout >= 256) return launch_tile<1, 1, 256>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
        if (ksize == 3 && stride == 1 && Cin % 4 == 0) return launch_tile<3, 1, 128>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
        if (ksize == 1 && stride == 1) return launch_tile<1, 1, 128>(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, pad, relu, st);
    }
    const int64_t Ntot = B * Lout;
    const int64_t gx = (Ntot + BN - 1) / BN;
    const int gy = (Cout + GBM - 1) / GBM;
    LBM_REQUIRE(gx <= 0x7fffffffLL && gy <= 65535, LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    hipLaunchKernelGGL(conv1d_generic_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), 0, st, x, w, bias, y, Ntot, Cin,
                       Lin, ldx, Cout, Lout, ldy, ksize, stride, pad, relu);
    return lbm::check_launch("lbm_conv1d_f32");
}

extern "C" int lbm_conv1d_f32(const float* x, const float* w, const float* bias, float* y, int64_t B, int32_t Cin,
                              int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                              lbm_stream_t stream) {
    const int64_t Lout = (ksize >= 1 && stride >= 1) ? ((int64_t)Lin + 2 * pad - ksize) / stride + 1 : 0;
    return lbm_conv1d_pitched_f32(x, Lin, w, bias, y, Lout, B, Cin, Lin, Cout, ksize, stride, pad, relu, stream);
}

// ---- Winograd layer with pre-transformed weights (caller-owned table)
extern "C" size_t lbm_conv1d_wino3_tables_bytes(int32_t Cin, int32_t Cout) {
    if (Cin < 8 || Cin % 8 != 0 || Cout < 64) return 0;
    return (size_t)(Cin / 8) * ((Cout + WinoCfg::BM - 1) / WinoCfg::BM) * 4 * kThreads * sizeof(float4);
}

extern "C" int lbm_conv1d_wino3_tables_init(void* tables, const float* w, int32_t Cin, int32_t Cout, lbm_stream_t stream) {
    LBM_REQUIRE(tables && w, LBM_EINVAL, "lbm_conv1d_wino3_tables_init: NULL pointer");
    LBM_REQUIRE(lbm_conv1d_wino3_tables_bytes(Cin, Cout) != 0, LBM_EUNSUPPORTED,
                "lbm_conv1d_wino3_tables_init: Cin=%d must be a multiple of 8, Cout=%d at least 64", Cin, Cout);
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_conv1d_wino3_tables_init: tables must be 16-byte aligned");
    const int mtiles = (Cout + WinoCfg::BM - 1) / WinoCfg::BM;
    hipLaunchKernelGGL(wino3_table_kernel, dim3(Cin / 8, mtiles), dim3(256), 0, lbm::as_stream(stream), w, static_cast<float*>(tables), Cin,
                       Cout, mtiles);
    return lbm::check_launch("lbm_conv1d_wino3_tables_init");
}

extern "C" int lbm_conv1d_wino3_f32(const float* x, int64_t ldx, const void* tables, const float* bias, float* y, int64_t ldy, int64_t B,
                                    int32_t Cin, int32_t Lin, int32_t Cout, int32_t relu, lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Lin >= 1, LBM_EINVAL, "lbm_conv1d_wino3_f32: bad sizes");
    LBM_REQUIRE(lbm_conv1d_wino3_tables_bytes(Cin, Cout) != 0, LBM_EUNSUPPORTED,
                "lbm_conv1d_wino3_f32: Cin=%d must be a multiple of 8, Cout=%d at least 64", Cin, Cout);
    const int Lout = Lin;   // ksize 3, stride 1, padding 1
    LBM_REQUIRE(ldx >= Lin && ldy >= Lout, LBM_EINVAL, "lbm_conv1d_wino3_f32: row pitch smaller than the row");
    LBM_REQUIRE((int64_t)Cin * ldx <= 0x7fffffffLL && (int64_t)Cout * ldy <= 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_conv1d_wino3_f32: one epoch exceeds 2^31 elements");
    if (B == 0) return LBM_OK;
    LBM_REQUIRE(x && tables && y, LBM_EINVAL, "lbm_conv1d_wino3_f32: NULL pointer");
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_conv1d_wino3_f32: tables must be 16-byte aligned");
    // Cin = 128 with 16-byte rows (the encoder's layer): the weights-stationary form (conv_wino_ws.hip), bit-identical results.
    // LBM_WINO3_KERNEL=tile (read once) keeps the tile kernel for A/B measurements.
    static const bool ws_off = [] { const char* v = getenv("LBM_WINO3_KERNEL"); return v != nullptr && strcmp(v, "tile") == 0; }();
    if (!ws_off && lbm::wino3_ws_eligible(x, ldx, y, ldy, Cin, Cout))
        return lbm::launch_wino3_ws(x, static_cast<const float*>(tables), bias, y, B, Lin, ldx, Cout, ldy, relu, lbm::as_stream(stream));
    return launch_wino3(x, static_cast<const float*>(tables), bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, relu, lbm::as_stream(stream), true);
}

// nn.Conv1d(kernel 4, stride 2, padding 1) as two 2-tap phases in the minimal form F(3,2): conv_k4s2_wino_ws.hip
extern "C" int lbm_conv1d_k4s2_wino_supported(int32_t Cin, int32_t Cout) { return Cin >= 1 && Cin <= 64 && Cout >= 64 && Cout % 64 == 0; }

extern "C" int lbm_conv1d_k4s2_wino_f32(const float* x, int64_t ldx, const float* w, const float* bias, float* y, int64_t ldy, int64_t B,
                                        int32_t Cin, int32_t Lin, int32_t Cout, int32_t relu, lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Lin >= 2, LBM_EINVAL, "lbm_conv1d_k4s2_wino_f32: bad sizes (B=%lld, Lin=%d)", (long long)B, Lin);
    LBM_REQUIRE(lbm_conv1d_k4s2_wino_supported(Cin, Cout), LBM_EUNSUPPORTED,
                "lbm_conv1d_k4s2_wino_f32: Cin=%d must be 1..64 and Cout=%d a multiple of 64 (use lbm_conv1d_pitched_f32)", Cin, Cout);
    const int Lout = (Lin + 2 - 4) / 2 + 1;
    LBM_REQUIRE(ldx >= Lin && ldy >= Lout, LBM_EINVAL, "lbm_conv1d_k4s2_wino_f32: row pitch smaller than the row");
    LBM_REQUIRE((int64_t)Cin * ldx <= 0x7fffffffLL && (int64_t)Cout * ldy <= 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_conv1d_k4s2_wino_f32: one epoch exceeds 2^31 elements");
    if (B == 0) return LBM_OK;
    LBM_REQUIRE(x && w && y, LBM_EINVAL, "lbm_conv1d_k4s2_wino_f32: NULL pointer");
    LBM_REQUIRE(lbm::k4s2_wino_eligible(x, ldx, w, y, ldy, Cin, Cout, Lin), LBM_EUNSUPPORTED,
                "lbm_conv1d_k4s2_wino_f32: rows must be 16-byte pitched (ldx, ldy multiples of 4) and x, w, y 16-byte aligned");
    return lbm::launch_k4s2_wino(x, w, bias, y, B, Cin, Lin, ldx, Cout, Lout, ldy, relu, lbm::as_stream(stream));
}
