// nn.Conv1d (+ReLU) forward as an implicit GEMM on the bf16 matrix cores with fp32-grade products (three bf16 pieces
// per operand, six piece products, fp32 accumulate -- the arithmetic of vq_bf16.hip, same accuracy against float64 as
// the fp32 fmaf chain).  Option `conv_method="bf16x6"` of the encoder (training/model.py:169-177); the default
// fp32-MFMA kernels of conv.hip stay the reference path.
//
// GEMM view: rows = output positions (b, p), K = (tap, input channel) with the channel count padded to a multiple
// of 16 so that a 16-wide k stage stays inside one tap, columns = output channels.  Workgroup = 128 positions x 256
// channels (one super-tile), 4 waves as 2 position groups x 2 channel groups; per stage the weights' pre-split
// slab (caller-owned table, lbm_conv1d_split_tables_init) goes global -> LDS by DMA and the 128 x 16 slice of the
// input window is split on the fly; 48 MFMAs per wave per stage, one barrier per stage.
#include "bf16x6.h"

namespace {

struct ConvGeom {
    int Cin, CinP, Lin, Lout, Cout, ksize, stride, pad, relu;
    int ldx, ldy;
};

// tab: [nst][3][CoutP][16] bf16, k = tap * CinP + ci, zero beyond (Cout, Cin); halves in LDS image order
__global__ __launch_bounds__(256) void conv_split_tables_kernel(const float* __restrict__ w, bf16x8* __restrict__ tab, int Cin, int CinP,
                                                                int Cout, int CoutP, int ksize, int nst) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // (stage, channel, half)
    if (i >= (int64_t)nst * CoutP * 2) return;
    const int half = (int)(i & 1);
    const int co = (int)((i >> 1) % CoutP);
    const int st = (int)((i >> 1) / CoutP);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = st * kKC + half * 8 + e;
        const int tap = k / CinP, ci = k - tap * CinP;
        v[e] = (co < Cout && ci < Cin && tap < ksize) ? w[((int64_t)co * Cin + ci) * ksize + tap] : 0.f;
    }
    bf16x8 p[3];
    split8(v, p[0], p[1], p[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) tab[(((int64_t)st * 3 + q) * CoutP + co) * 2 + (half ^ ((co >> 3) & 1))] = p[q];
}

__global__ __launch_bounds__(kThreads, 2) void conv_bf16x6_kernel(const float* __restrict__ x, const u32x4* __restrict__ tab,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int N,
                                                                  ConvGeom g, int CoutP, int nst) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* const s_bias = reinterpret_cast<float*>(smem + 2 * kStageBytes);   // [2][kSup]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rg = wave & 1, cg = wave >> 1;
    const int nsup = CoutP / kSup;
    const int total = nsup * nst;

    // x: thread -> (row = tid / 2, octet = tid & 1) of the stage's 128 x 16 slice
    const int xr = tid >> 1, xo = tid & 1;
    unsigned n = blockIdx.x * kRows + xr;
    n = n < (unsigned)N ? n : (unsigned)N - 1;
    const unsigned xb_i = n / (unsigned)g.Lout;
    const int xp0 = (int)(n - xb_i * g.Lout) * g.stride - g.pad;       // input position of tap 0
    const float* xb = x + (int64_t)xb_i * g.Cin * g.ldx;
    float xv[8];
    const int tstep = 3 * CoutP * 2;

    // cursor of the stage loaded last: super-tile, stage, and the stage's k as (tap, first channel); wave-uniform
    int rsup = 0, rst = 0, rtap = 0, rci = 0;
    auto load_stage = [&]() {
        const int tap = rtap, ci0 = rci + xo * 8;
        int pos = xp0 + tap;
        pos = pos < 0 ? 0 : (pos < g.Lin ? pos : g.Lin - 1);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ci = ci0 + e < g.Cin ? ci0 + e : g.Cin - 1;
            xv[e] = xb[ci * g.ldx + pos];
        }
    };
    auto copy_codes = [&](int sup, int st, int buf) {
        const u32x4* tp = tab + (st * tstep + sup * (kSup * 2) + tid);
        unsigned char* sb = smem + buf * kStageBytes + tid * 16;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(tp + (i >> 1) * (CoutP * 2) + (i & 1) * kThreads),
                                             (__attribute__((address_space(3))) void*)(sb + (i >> 1) * kCodePlane + (i & 1) * (kThreads * 16)), 16, 0, 0);
    };
    auto store_stage = [&](int buf) {   // the stage loaded last
        const int tap = rtap, ci0 = rci + xo * 8;
        const int pos = xp0 + tap;
        const bool inside = pos >= 0 && pos < g.Lin;     // zero padding of the convolution
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            asm volatile("" : "+v"(xv[e]));
            v[e] = (inside && ci0 + e < g.Cin) ? xv[e] : 0.f;
        }
        bf16x8 p1, p2, p3;
        split8(v, p1, p2, p3);
        unsigned char* xbuf = smem + buf * kStageBytes + 3 * kCodePlane + swz(xr, xo);
        *reinterpret_cast<bf16x8*>(xbuf) = p1;
        *reinterpret_cast<bf16x8*>(xbuf + kXPlane) = p2;
        *reinterpret_cast<bf16x8*>(xbuf + 2 * kXPlane) = p3;
    };
    auto stage_bias = [&](int sup) {   // bias of the super-tile, read back by the epilogue after >= one barrier
        const int co = sup * kSup + tid;
        s_bias[(sup & 1) * kSup + tid] = (bias != nullptr && co < g.Cout) ? bias[co] : 0.f;
    };

    f32x16 acc[2][4];   // [position tile][channel tile]
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    const int a_off = swz(cg * 128 + j, h), b_off = 3 * kCodePlane + swz(rg * 64 + j, h);
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

    const float lo = g.relu ? 0.f : -INFINITY;

    auto advance = [&]() {
        rci += kKC;
        if (rci == g.CinP) {
            rci = 0;
            ++rtap;
        }
        if (++rst == nst) {
            rst = 0;
            rtap = 0;
            ++rsup;
        }
    };
    load_stage();
    copy_codes(0, 0, 0);
    stage_bias(0);
    store_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (total > 1) {
        advance();
        load_stage();
    }
    int gi = 0;
    for (int sup = 0; sup < nsup; ++sup) {
        for (int st = 0; st < nst; ++st, ++gi) {
            const int buf = gi & 1;
            if (gi + 1 < total) {
                store_stage(buf ^ 1);
                copy_codes(rsup, rst, buf ^ 1);
                if (rst == 0) stage_bias(rsup);
                if (gi + 2 < total) {
                    advance();
                    load_stage();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_stage(buf);
            if (st == nst - 1) {   // the super-tile's 128 x 256 outputs: bias, ReLU, store (lanes = consecutive positions)
                const float* sb_ = s_bias + (sup & 1) * kSup + cg * 128;
                const int co0 = sup * kSup + cg * 128;
                const bool full = co0 + 128 <= g.Cout && (blockIdx.x + 1) * kRows <= (unsigned)N;   // uniform per wave
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const unsigned row = blockIdx.x * kRows + rg * 64 + rt * 32 + j;
                    const bool rok = row < (unsigned)N;
                    const unsigned rc = rok ? row : 0u;
                    const unsigned b = rc / (unsigned)g.Lout;
                    float* yp = y + ((int64_t)b * g.Cout + co0 + 4 * h) * g.ldy + (rc - b * g.Lout);
                    asm volatile("" : "+v"(yp));   // keeps the 64 store addresses from being hoisted out of the k loop
                    __attribute__((address_space(1))) float* const yg = (__attribute__((address_space(1))) float*)yp;
                    const float* sbl = sb_ + 4 * h;
                    if (full) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int cl = ct * 32 + (r & 3) + 8 * (r >> 2);   // + 4 h: in yp, sbl
                                yg[cl * g.ldy] = fmaxf(acc[rt][ct][r] + sbl[cl], lo);
                            }
                    } else {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int cl = ct * 32 + (r & 3) + 8 * (r >> 2);
                                if (rok && co0 + 4 * h + cl < g.Cout) yg[cl * g.ldy] = fmaxf(acc[rt][ct][r] + sbl[cl], lo);
                            }
                    }
                }
                zero_acc();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
}

struct ConvTabPlan {
    int CinP, CoutP, nst;
    size_t bytes;
};
ConvTabPlan conv_tab_plan(int Cin, int Cout, int ksize) {
    ConvTabPlan p;
    p.CinP = (Cin + kKC - 1) / kKC * kKC;
    p.CoutP = (Cout + kSup - 1) / kSup * kSup;
    p.nst = p.CinP * ksize / kKC;
    p.bytes = (size_t)p.nst * 3 * p.CoutP * 32;
    return p;
}

}  // namespace

extern "C" size_t lbm_conv1d_split_tables_bytes(int32_t Cin, int32_t Cout, int32_t ksize) {
    if (Cin < 1 || Cout < 1 || ksize < 1 || ksize > 64 || Cin > 4096 || Cout > 65536) return 0;
    return conv_tab_plan(Cin, Cout, ksize).bytes;
}

extern "C" int lbm_conv1d_split_tables_init(void* tables, const float* w, int32_t Cin, int32_t Cout, int32_t ksize, lbm_stream_t stream) {
    LBM_REQUIRE(tables && w, LBM_EINVAL, "lbm_conv1d_split_tables_init: NULL pointer");
    LBM_REQUIRE(lbm_conv1d_split_tables_bytes(Cin, Cout, ksize) != 0, LBM_EUNSUPPORTED,
                "lbm_conv1d_split_tables_init: unsupported shape (Cin=%d, Cout=%d, ksize=%d)", Cin, Cout, ksize);
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_conv1d_split_tables_init: tables must be 16-byte aligned");
    const ConvTabPlan p = conv_tab_plan(Cin, Cout, ksize);
    const int64_t items = (int64_t)p.nst * p.CoutP * 2;
    hipLaunchKernelGGL(conv_split_tables_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, lbm::as_stream(stream), w,
                       static_cast<bf16x8*>(tables), Cin, p.CinP, Cout, p.CoutP, ksize, p.nst);
    return lbm::check_launch("lbm_conv1d_split_tables_init");
}

extern "C" int lbm_conv1d_bf16x6_f32(const float* x, int64_t ldx, const void* tables, const float* bias, float* y, int64_t ldy,
                                     int64_t B, int32_t Cin, int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad,
                                     int32_t relu, lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && ksize >= 1 && stride >= 1 && pad >= 0 && Lin >= 0, LBM_EINVAL,
                "lbm_conv1d_bf16x6_f32: bad shape");
    LBM_REQUIRE(lbm_conv1d_split_tables_bytes(Cin, Cout, ksize) != 0, LBM_EUNSUPPORTED, "lbm_conv1d_bf16x6_f32: unsupported shape");
    const int64_t Lout64 = Lin + 2 * (int64_t)pad - ksize >= 0 ? (Lin + 2 * (int64_t)pad - ksize) / stride + 1 : 0;
    if (B == 0 || Lout64 == 0) return LBM_OK;
    LBM_REQUIRE(x && tables && y, LBM_EINVAL, "lbm_conv1d_bf16x6_f32: NULL pointer");
    LBM_REQUIRE(ldx >= Lin && ldy >= Lout64, LBM_EINVAL, "lbm_conv1d_bf16x6_f32: row pitch shorter than the row");
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_conv1d_bf16x6_f32: tables must be 16-byte aligned");
    LBM_REQUIRE((int64_t)Cin * ldx < 0x7fffffffLL && ldy < 0x7fffffffLL && Lout64 < 0x7fffffffLL, LBM_EUNSUPPORTED,
                "lbm_conv1d_bf16x6_f32: Cin * ldx must be below 2^31");
    const ConvTabPlan p = conv_tab_plan(Cin, Cout, ksize);
    LBM_REQUIRE(p.bytes / 16 < 0x7fffffffULL, LBM_EUNSUPPORTED, "lbm_conv1d_bf16x6_f32: table too large");
    const int64_t N = B * Lout64;
    const int64_t blocks = (N + kRows - 1) / kRows;
    LBM_REQUIRE(N < 0x7fffff00LL, LBM_EUNSUPPORTED, "lbm_conv1d_bf16x6_f32: B * Lout must be below 2^31");
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_conv1d_bf16x6_f32: grid too large");
    ConvGeom g;
    g.Cin = Cin; g.CinP = p.CinP; g.Lin = Lin; g.Lout = (int)Lout64; g.Cout = Cout; g.ksize = ksize; g.stride = stride; g.pad = pad;
    g.relu = relu; g.ldx = (int)ldx; g.ldy = (int)ldy;
    const int lds = 2 * kStageBytes + 2 * kSup * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bf16x6_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_bf16x6_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(conv_bf16x6_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, lbm::as_stream(stream), x,
                       static_cast<const u32x4*>(tables), bias, y, (int)N, g, p.CoutP, p.nst);
    return lbm::check_launch("lbm_conv1d_bf16x6_f32");
}
