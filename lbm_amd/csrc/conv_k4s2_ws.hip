// lbm_amd -- the first two encoder layers (Conv1d 30 -> 64 and 64 -> 128, 4 taps, stride 2, padding 1, + ReLU; reference:
// training/model.py:170-172) as WEIGHTS-STATIONARY kernels (round 3), the direct-form sibling of conv_wino_ws.hip.
//
// Same products in the same order as conv_tile_kernel<4, 2, ...> (conv.hip) -- chunks of 8 input channels ascending, inside a chunk the
// two lane halves own the first / second four channels, taps ascending; a ragged last chunk shifted down to end at Cin with the weights of
// the channels it shares with the chunk before zeroed -- so the results are that kernel's bit for bit
// (tests: test_conv1d_k4s2_ws_is_the_tile_kernel_bit_for_bit).
//
// A wave owns 32 output channels and 64 of the tile's output positions for the whole launch: its weights W[32 ch][Cin x 4] are its MFMA B
// operand in 64 / 128 registers, loaded once.  Workgroups (NCG channel groups x NPG position groups = 8 waves = ALL output channels x 64 NPG
// positions, one per CU) are persistent and walk the tiles; per tile only the raw input window moves -- Cin rows x (128 NPG + 8) samples,
// 61 / 66 KiB -- dropped into LDS by LDS-DMA under the previous tile's MFMA stream (two buffers).  ONE barrier per tile (128 / 256 MFMAs of
// a wave); no staging registers, no ds_write, no arithmetic on the operands: row r of position sub-tile pm is output position 2 r + pm, so
// lane j's two positions need the window slots 4 j + 3 .. 4 j + 8 of an input-channel row = one aligned 16-byte read + two 8-byte reads per
// eight MFMAs (four taps x two sub-tiles).  With two accumulator sets the bias / ReLU / stores of tile t (32 contiguous bytes per lane
// and register quad) ride under the MFMA stream of tile t + 1.
//
// What it replaces (profiles/r03_experiments.md, stamps of conv_tile_kernel): a tile of the first layer spent 48 % of its time in its
// prologue and epilogue (one exposed load round trip, the LDS writes, 16 scattered stores per lane with their drain) and 22 % in the
// chunk loop's staging.
// (NCG x NPG = 4 waves, two such workgroups per CU with their own barriers, was measured: conv1 2.03 vs 2.08 ms, conv2 3.47 vs 3.44 --
// independent workgroups do not buy what the lockstep of one was thought to cost; the 8-wave shapes stay.)
#include <type_traits>

#include "common.h"
#include "conv_ws.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));


template <int CIN, int NCG, int NPG>
struct K4Cfg {
    static constexpr int NW = NCG * NPG;                    // waves per workgroup: channel groups of 32 x position groups of 64
    static constexpr int TP = 64 * NPG;                     // output positions per tile
    static constexpr int XW = 2 * TP + 8;                   // window samples per input-channel row: samples 2 p0 - 4 .. 2 p0 + 2 TP + 3
    static constexpr int X4 = XW / 4;
    static constexpr int NCH = (CIN + 7) / 8;               // chunks of 8 input channels (the tile kernel's)
    static constexpr int KSTEPS = 16 * NCH;                 // MFMA k-steps per tile and position sub-tile
    static constexpr int NRS = 4 * NCH;                     // row steps: one input-channel row per lane half = 4 taps
    static constexpr int QUADS = CIN * X4;
    static constexpr int PIECES = (QUADS + 63) / 64;        // 1 KiB DMA pieces per window
    static constexpr int BUF_FLOATS = PIECES * 256;
    static constexpr int SLOTS = (PIECES + NW - 1) / NW;    // pieces per wave
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_FLOATS * sizeof(float);
    static_assert(CIN >= 8 && LDS_BYTES <= 160 * 1024, "window does not fit");
    static_assert((CIN - 5) * XW * 4 + 64 < 65536, "row offsets should fit the 16-bit immediate of a DS instruction");
    static_assert(SLOTS <= NRS && 8 <= NRS, "the side-work schedule needs a row step per request and per store unit");
};

__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_dst) {
    unsigned keep;
    // (readfirstlane: under scalar-register pressure hipcc has handed this wave-uniform value to the "s" operand in a VGPR)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}

// Source of the window quads that lie wholly outside [0, Lin) (the convolution's zero padding, everything right of a row's end) and of the
// filler quads behind the last row
__device__ __align__(16) float g_zero_quad_k4[4];

template <int CIN, int NCG, int NPG>
__global__ __launch_bounds__(64 * NCG * NPG) void conv_k4s2_ws_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, int ldx, int ldy,
                                                                int Lin, int Lout, int Cout, int relu, int ptiles, int64_t ntiles) {
    using Cfg = K4Cfg<CIN, NCG, NPG>;
    constexpr int NW = Cfg::NW, TP = Cfg::TP, XW = Cfg::XW, X4 = Cfg::X4, NCH = Cfg::NCH, NRS = Cfg::NRS, SLOTS = Cfg::SLOTS;
    extern __shared__ __align__(16) float smem[];              // 2 x [CIN rows][XW] (+ filler up to a whole piece)
    const unsigned lds_x = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)smem;

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int cgi = wave % NCG, pgi = wave / NCG;              // this wave's 32 channels and its 64 positions of the tile
    const int co = cgi * 32 + j;                               // (Cout = 32 NCG: checked by the launcher)

    // ---- B operand: k-step t = 16 ch + 4 cl + kt of lane half h is input channel ci0(ch) + 4 h + cl, tap kt
    float breg[Cfg::KSTEPS];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int ci0 = 8 * ch + 8 <= CIN ? 8 * ch : CIN - 8;  // a ragged last chunk is shifted down ...
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int ci = ci0 + 4 * h + cl;
            f4 v = *reinterpret_cast<const f4*>(w + ((int64_t)co * CIN + ci) * 4);
            if (ci < 8 * ch) v = f4{0.f, 0.f, 0.f, 0.f};       // ... and the channels the chunk before already covered are zeroed
            breg[16 * ch + 4 * cl] = v.x; breg[16 * ch + 4 * cl + 1] = v.y; breg[16 * ch + 4 * cl + 2] = v.z; breg[16 * ch + 4 * cl + 3] = v.w;
        }
    }
    const float bv = bias != nullptr ? bias[co] : 0.f;
    const float lo = relu ? 0.f : -INFINITY;

    const int npieces = wave < Cfg::PIECES - NW * (SLOTS - 1) ? SLOTS : SLOTS - 1;
    // Requests piece k of this wave (window quads 64 (wave + 8 k) ..) of tile (b, pt) into buffer `buf`.  Quads wholly outside [0, Lin)
    // come from g_zero_quad_k4, so only the one quad per row that straddles the row's end needs fixing afterwards.
    const float* vzero = g_zero_quad_k4;   // (in a register pair: fetched per request it is a scalar load through the GOT + an lgkmcnt(0) wait)
    asm volatile("" : "+v"(vzero));
    auto request = [&](int64_t b, int pt, int buf, int k) {
        if (k >= npieces) return;                               // wave-uniform
        const int q = 64 * (wave + NW * k) + lane;
        const int row = q / X4, c4 = q - row * X4;
        const int g = 2 * pt * TP - 4 + 4 * c4;
        const float* src = x + b * (int64_t)CIN * ldx;
        const float* a = (q >= Cfg::QUADS || g < 0 || g >= Lin) ? vzero : src + (row * ldx + g);
        lds_dma16(a, lds_x + (unsigned)(buf * Cfg::BUF_FLOATS + 256 * (wave + NW * k)) * 4u);
    };
    auto fix_edges = [&](int pt, int buf) {
        const int ga = 2 * pt * TP - 4;
        if ((Lin & 3) == 0 || ga + XW <= Lin || ga >= Lin) return;   // wave-uniform
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {
            if (k < npieces) {
                const int q = 64 * (wave + NW * k) + lane;
                const int row = q / X4, c4 = q - row * X4;
                const int g = ga + 4 * c4;
                if (q < Cfg::QUADS && g < Lin && g + 4 > Lin) {
                    float* d = smem + buf * Cfg::BUF_FLOATS + 256 * (wave + NW * k) + 4 * lane;
#pragma unroll
                    for (int e = 1; e < 4; ++e)
                        if (g + e >= Lin) d[e] = 0.f;
                }
            }
        }
    };

    f32x16 acc[2][2];   // [tile parity][position sub-tile]
    // One tile: NRS row steps of 8 MFMAs; the reads of row step r + 1 are requested before row step r's MFMAs are issued.
    auto mfma_tile = [&](auto par_tag, auto&& hook) {
        constexpr int P = decltype(par_tag)::value;
        typedef const volatile __attribute__((address_space(3))) f4* lds_vf4;
        typedef const volatile __attribute__((address_space(3))) f2* lds_vf2;
        const float* base = smem + P * Cfg::BUF_FLOATS + 4 * h * XW + pgi * 128 + 4 * j;
        const lds_vf4 xq = (lds_vf4)(base + 4);
        const lds_vf2 xl = (lds_vf2)(base + 2), xr = (lds_vf2)(base + 8);
        f4 tq[2];
        f2 tl[2], tr[2];
        auto fetch = [&](int rs, int set) {
            const int ch = rs / 4, cl = rs % 4;
            const int ci0 = 8 * ch + 8 <= CIN ? 8 * ch : CIN - 8;
            const int off = (ci0 + cl) * XW;
            tq[set] = xq[off / 4];
            tl[set] = xl[off / 2];
            tr[set] = xr[off / 2];
        };
        fetch(0, 0);
#pragma unroll
        for (int rs = 0; rs < NRS; ++rs) {
            if (rs + 1 < NRS) fetch(rs + 1, (rs + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            const f4 q = tq[rs & 1];
            const float a0[4] = {tl[rs & 1].y, q.x, q.y, q.z};      // position 2 r:     window slots 4 j + 3 .. 4 j + 6
            const float a1[4] = {q.y, q.z, q.w, tr[rs & 1].x};      // position 2 r + 1: window slots 4 j + 5 .. 4 j + 8
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const int t = 4 * rs + kt;
                if (t == 0) {
                    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kt], breg[0], z, 0, 0, 0);
                    acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kt], breg[0], z, 0, 0, 0);
                } else {
                    acc[P][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kt], breg[t], acc[P][0], 0, 0, 0);
                    acc[P][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kt], breg[t], acc[P][1], 0, 0, 0);
                }
            }
            hook(rs);   // side work issued while these MFMAs occupy the pipe: window requests, the previous tile's outputs
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Output unit u of 8 = register quad rq = u >> 1, rows 8 rq + 4 h + 2 (u & 1) + (0, 1): row r of sub-tile pm is position 2 r + pm, so a
    // unit is four consecutive positions, a register quad eight: 32 contiguous bytes per lane.
    auto output = [&](auto par_tag, int64_t b, int pt, int u) {
        constexpr int P = decltype(par_tag)::value;
        const int rq = u >> 1, e = 2 * (u & 1);
        f4 o;
        o.x = fmaxf(acc[P][0][4 * rq + e] + bv, lo);
        o.y = fmaxf(acc[P][1][4 * rq + e] + bv, lo);
        o.z = fmaxf(acc[P][0][4 * rq + e + 1] + bv, lo);
        o.w = fmaxf(acc[P][1][4 * rq + e + 1] + bv, lo);
        const int p = pt * TP + pgi * 64 + 2 * (8 * rq + 4 * h + e);
        // ldy is a multiple of 4 >= Lout: the tail quad may spill into the row's pitch padding (as conv_tile_kernel's does)
        if (p < Lout) *reinterpret_cast<f4*>(y + (b * Cout + co) * (int64_t)ldy + p) = o;
    };

    // ---- the tile walk: tile = block, block + grid, ...; (b, pt) kept incrementally
    int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;                        // workgroup-uniform
    const int nstreams = gridDim.x;
    int64_t b = tile / ptiles;
    int pt = (int)(tile - b * ptiles);
    const int db = nstreams / ptiles, dpt = nstreams - db * ptiles;
    int64_t pb = 0;
    int ppt = 0;
    bool have_prev = false;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) request(b, pt, 0, k);

    auto tile_step = [&](auto par_tag) -> bool {
        constexpr int P = decltype(par_tag)::value;
        const std::integral_constant<int, 1 - P> other{};
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        fix_edges(pt, P);
        __syncthreads();           // this tile's window is in buffer P; every wave is done with buffer 1 - P
        int64_t nb = b + db;
        int npt = pt + dpt;
        if (npt >= ptiles) { npt -= ptiles; ++nb; }
        const int64_t ntile = tile + nstreams;
        const bool more = ntile < ntiles;
        mfma_tile(par_tag, [&](int rs) {
            // both kinds of side work go out in the first row steps: the tile's closing s_waitcnt vmcnt(0) covers the stores as well,
            // so a store issued late would be waited out there
            if (rs < 8 && have_prev) output(other, pb, ppt, rs);
            if (rs < SLOTS && more) request(nb, npt, 1 - P, rs);
        });
        pb = b; ppt = pt; have_prev = true;
        if (more) {
            tile = ntile; b = nb; pt = npt;
            return false;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) output(par_tag, pb, ppt, u);
        return true;
    };
    while (true) {
        if (tile_step(std::integral_constant<int, 0>{})) break;
        if (tile_step(std::integral_constant<int, 1>{})) break;
    }
}

int g_cus[64];
bool g_attr[64][2];

template <int CIN, int NCG, int NPG>
int launch(int which, const float* x, const float* w, const float* bias, float* y, int64_t B, int Lin, int64_t ldx, int Lout, int64_t ldy,
           int relu, hipStream_t st) {
    using Cfg = K4Cfg<CIN, NCG, NPG>;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_f32: hipGetDevice failed");
    if (g_cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus[dev] = n;
    }
    auto k = conv_k4s2_ws_kernel<CIN, NCG, NPG>;
    if (!g_attr[dev][which]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
        g_attr[dev][which] = true;
    }
    const int ptiles = (Lout + Cfg::TP - 1) / Cfg::TP;
    const int64_t ntiles = B * ptiles;
    const int64_t slots = (int64_t)g_cus[dev] * (Cfg::NW <= 4 ? 2 : 1);  // resident workgroups: one 8-wave or two 4-wave per CU (LDS, registers)
    const int grid = (int)(ntiles < slots ? ntiles : slots);            // each walks ntiles / grid tiles
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(64 * Cfg::NW), Cfg::LDS_BYTES, st, x, w, bias, y, (int)ldx, (int)ldy, Lin, Lout, 32 * NCG, relu,
                       ptiles, ntiles);
    return lbm::check_launch("lbm_conv1d_f32");
}

}  // namespace

namespace lbm {

bool k4s2_ws_eligible(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, int Cin, int Cout, int Lin) {
    return ((Cin == 30 && Cout == 64) || (Cin == 64 && Cout == 128)) && Lin >= 2 && ldx % 4 == 0 && ldy % 4 == 0 &&
           ((((uintptr_t)x) | ((uintptr_t)y) | ((uintptr_t)w)) & 15) == 0;
}

int launch_k4s2_ws(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout, int Lout,
                   int64_t ldy, int relu, hipStream_t st) {
    if (Cin == 30) return launch<30, 2, 4>(0, x, w, bias, y, B, Lin, ldx, Lout, ldy, relu, st);
    return launch<64, 4, 2>(1, x, w, bias, y, B, Lin, ldx, Lout, ldy, relu, st);
}

}  // namespace lbm
