// Conv1d (+bias, +ReLU) of the VQ-VAE encoder (a10) as an implicit GEMM on fp32 MFMA.
//
// y[b, co, p] = bias[co] + sum_{ci,kt} w[co, ci, kt] * x[b, ci, p*stride - pad + kt]
// GEMM view per epoch: M = Cout, N = Lout, K = Cin*ksize; A = w viewed as [Cout][Cin*ksize] (torch's
// weight layout already is that row-major matrix).  im2col is never materialised.
//
// conv_tile_kernel<KS, STRIDE, BM> (the encoder's four layer shapes):
//   workgroup tile BM (co) x 128 (positions of ONE epoch), 4 wavefronts as 2x2, v_mfma_f32_32x32x2_f32.
//   K is consumed CK input channels at a time.  Per chunk the LDS holds the weight slab [BM][CK*KS(+4)]
//   and the raw input rows [CK][(128-1)*STRIDE+KS] -- the MFMA B operand for tap kt of channel ci is
//   read straight from the input row at offset p*STRIDE+kt, so the KS-fold im2col replication never
//   exists, global loads are contiguous row segments, and both LDS images are double buffered with the
//   next chunk's global loads in flight (register staged) while the current chunk's MFMAs run.
// conv1d_generic_kernel: any ksize/stride/pad (runtime), simple single-buffered gather; used for shapes
//   outside the instantiations above.
// fp32 MFMA = exact fp32 fmaf chain; summation order: chunks ascending, inside a chunk the two lane
// halves own the first/second half of the chunk's (ci, kt) list in ascending order.
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kThreads = 256;
constexpr int BN = 128;

template <int KS, int STRIDE, int BM>
struct ConvCfg {
    static constexpr int CK = (KS == 1) ? 32 : 8;        // input channels per chunk
    static constexpr int KC = CK * KS;                   // GEMM-K per chunk (32 / 24 / 32)
    static constexpr int LDA = KC + 4;                   // padded weight-slab row (conflict-free ds_read_b128)
    static constexpr int XW = (BN - 1) * STRIDE + KS;    // input samples needed per channel row
    static constexpr int XWP = (XW + 3) & ~3;
    static constexpr int A_F4 = BM * KC / 4 / kThreads;  // float4 per thread per chunk (weights)
    static constexpr int X_IT = (CK * XW + kThreads - 1) / kThreads;
    static constexpr int MS = BM / 64;                   // 32-row MFMA tiles per wave along M (wave tile = BM/2 x 64)
    static constexpr size_t LDS_FLOATS = 2 * (size_t)(BM * LDA + CK * XWP) + BM;   // + bias tile
    static_assert(KC % 8 == 0, "each lane half consumes KC/2 k-steps in groups of 4");
    static_assert((BM * KC / 4) % kThreads == 0, "weight slab must split evenly");
};

template <int KS, int STRIDE, int BM>
__global__ __launch_bounds__(kThreads) void conv_tile_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             int Cin, int Lin, int Cout, int Lout, int pad, int relu,
                                                             int ptiles, int mtiles) {
    using Cfg = ConvCfg<KS, STRIDE, BM>;
    constexpr int CK = Cfg::CK, KC = Cfg::KC, LDA = Cfg::LDA, XW = Cfg::XW, XWP = Cfg::XWP, MS = Cfg::MS;
    extern __shared__ __align__(16) float smem[];
    float* const s_a0 = smem;                 // 2 x [BM][LDA]
    float* const s_x0 = smem + 2 * BM * LDA;  // 2 x [CK][XWP]
    float* const s_bias = smem + Cfg::LDS_FLOATS - BM;   // [BM], visible after the first barrier of the chunk loop

    const int Ktot = Cin * KS;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    // block -> (m tile fastest, then position tile, then epoch): blocks sharing an input tile are neighbours
    const int mt = blockIdx.x % mtiles;
    const int rest = blockIdx.x / mtiles;
    const int pt = rest % ptiles;
    const int64_t b = rest / ptiles;
    const int m0 = mt * BM, p0 = pt * BN;
    const int g0 = p0 * STRIDE - pad;
    const float* xb = x + b * (int64_t)Cin * Lin;

    if (threadIdx.x < BM) {
        const int co = m0 + threadIdx.x;
        float bv = 0.f;
        if (bias != nullptr && co < Cout) bv = bias[co];
        s_bias[threadIdx.x] = bv;
    }

    float4 ra[Cfg::A_F4];
    float rx[Cfg::X_IT];

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
    int x_off[Cfg::X_IT], x_lds[Cfg::X_IT];
    bool x_ok[Cfg::X_IT];
#pragma unroll
    for (int i = 0; i < Cfg::X_IT; ++i) {
        const int f = threadIdx.x + i * kThreads;
        const int r = f / XW, c = f - r * XW;
        const int g = g0 + c;
        x_ok[i] = r < CK && g >= 0 && g < Lin;
        x_off[i] = (r < CK ? r : CK - 1) * Lin + (g < 0 ? 0 : (g < Lin ? g : Lin - 1));
        x_lds[i] = (r < CK ? r : CK - 1) * XWP + c;
    }

    // Chunk ch covers input channels [ch*CK, ch*CK+CK).  A ragged last chunk is shifted down to end at Cin
    // (needs Cin >= CK) and the weights of the channels it shares with the previous chunk are zeroed, so
    // every x row read is a real row.  Loads are unconditional and masked afterwards: a "load or zero"
    // select would make the compiler branch around every load and drain vmcnt per element.
    auto load_chunk = [&](int ch) {
        const int cn = ch * CK;
        const int ci0 = cn + CK <= Cin ? cn : Cin - CK;
        const float* wc = w + ci0 * KS;
        const float* xc = xb + (int64_t)ci0 * Lin;
#pragma unroll
        for (int i = 0; i < Cfg::A_F4; ++i) ra[i] = *reinterpret_cast<const float4*>(wc + a_off[i]);
#pragma unroll
        for (int i = 0; i < Cfg::X_IT; ++i) rx[i] = xc[x_off[i]];
    };
    // Writes the registers loaded by load_chunk(ch) to LDS image `buf`, masking here (not at the load): the
    // empty asm pins each loaded value so the compiler cannot sink a load back under its mask.
    auto store_chunk = [&](int buf, int ch) {
        const int cn = ch * CK;
        const int shift = (cn + CK <= Cin ? 0 : cn + CK - Cin) * KS;   // k entries the shifted chunk re-reads
        float* sa = s_a0 + buf * (BM * LDA);
        float* sx = s_x0 + buf * (CK * XWP);
#pragma unroll
        for (int i = 0; i < Cfg::A_F4; ++i) {
            asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
            const bool ok = a_ok[i] && a_k[i] >= shift;
            *reinterpret_cast<float4*>(sa + a_lds[i]) = ok ? ra[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < Cfg::X_IT; ++i) {
            asm volatile("" : "+v"(rx[i]));
            if (threadIdx.x + i * kThreads < CK * XW) sx[x_lds[i]] = x_ok[i] ? rx[i] : 0.f;
        }
    };

    f32x16 acc[MS][2];
#pragma unroll
    for (int a = 0; a < MS; ++a)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.f;

    // k-steps [4*qlo, 4*qhi) of this lane half's list, from LDS image `buf`
    auto mfma_block = [&](int buf, int qlo, int qhi) {
        const float* ap = s_a0 + buf * (BM * LDA) + (wm * (BM / 2) + j) * LDA + h * (KC / 2);
        const float* bp = s_x0 + buf * (CK * XWP) + (h * (CK / 2)) * XWP + (wn * 64 + j) * STRIDE;
#pragma unroll
        for (int q = qlo; q < qhi; ++q) {
            float av[MS][4];
#pragma unroll
            for (int a = 0; a < MS; ++a) {
                const float4 t = *reinterpret_cast<const float4*>(ap + a * 32 * LDA + 4 * q);
                av[a][0] = t.x; av[a][1] = t.y; av[a][2] = t.z; av[a][3] = t.w;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = 4 * q + u;            // k-step inside this lane half's list
                const int ci = s / KS, kt = s % KS;  // compile-time after unrolling
                const float b0 = bp[ci * XWP + kt];
                const float b1 = bp[ci * XWP + kt + 32 * STRIDE];
#pragma unroll
                for (int a = 0; a < MS; ++a) {
                    acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][u], b0, acc[a][0], 0, 0, 0);
                    acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[a][u], b1, acc[a][1], 0, 0, 0);
                }
            }
        }
    };

    // Software pipeline, one barrier per chunk: while chunk ch is multiplied out of LDS image ch&1, chunk
    // ch+1 (loaded one iteration ago, in registers) is written to the other image and chunk ch+2's global
    // loads are issued -- all in the middle of the MFMA stream, so staging runs in the MFMAs' shadow.
    constexpr int QN = KC / 8, QH = QN / 2;
    const int nchunks = (Cin + CK - 1) / CK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    load_chunk(nchunks > 1 ? 1 : 0);
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        const int buf = ch & 1;
        mfma_block(buf, 0, QH);
        store_chunk(buf ^ 1, ch + 1);
        load_chunk(ch + 2 < nchunks ? ch + 2 : nchunks - 1);   // the tail re-reads the last chunk (unused)
        mfma_block(buf, QH, QN);
        __syncthreads();
    }
    mfma_block((nchunks - 1) & 1, 0, QN);

    // Epilogue: bias comes from the LDS tile staged at kernel start (a global load per element would drain
    // vmcnt 16*MS*2 times); stores carry no waits.  Each store instruction writes two 128-byte row segments.
    float* yb = y + b * (int64_t)Cout * Lout;
    const float lo = relu ? 0.f : -INFINITY;
    const bool full_m = m0 + BM <= Cout;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int p = p0 + wn * 64 + n * 32 + j;
        if (p >= Lout) continue;
#pragma unroll
        for (int a = 0; a < MS; ++a) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int cl = wm * (BM / 2) + a * 32 + 8 * rq + 4 * h;
                const float4 bq = *reinterpret_cast<const float4*>(s_bias + cl);
                const float bv[4] = {bq.x, bq.y, bq.z, bq.w};
                float* yp = yb + (int64_t)(m0 + cl) * Lout + p;
                if (full_m) {
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri) yp[(int64_t)ri * Lout] = fmaxf(acc[a][n][4 * rq + ri] + bv[ri], lo);
                } else {
#pragma unroll
                    for (int ri = 0; ri < 4; ++ri)
                        if (m0 + cl + ri < Cout) yp[(int64_t)ri * Lout] = fmaxf(acc[a][n][4 * rq + ri] + bv[ri], lo);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- generic path
constexpr int GBM = 64, GKC = 32;
constexpr int GLDA = GKC + 4;

__global__ __launch_bounds__(kThreads) void conv1d_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int64_t Ntot, int Cin, int Lin, int Cout, int Lout, int ks,
                                                                  int stride, int pad, int relu) {
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
    const float* xcol = x + bcol * (int64_t)Cin * Lin;
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
                if (col_ok && ci < Cin && pos >= 0 && pos < Lin) v = xcol[(int64_t)ci * Lin + pos];
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
        float* yb = y + b * (int64_t)Cout * Lout + p;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < Cout) {
                float v = (sub == 0 ? acc0[r] : acc1[r]) + (bias != nullptr ? bias[co] : 0.f);
                if (relu) v = fmaxf(v, 0.f);
                yb[(int64_t)co * Lout] = v;
            }
        }
    }
}

template <int KS, int STRIDE, int BM>
int launch_tile(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int Cout, int Lout,
                int pad, int relu, hipStream_t st) {
    using Cfg = ConvCfg<KS, STRIDE, BM>;
    const int ptiles = (Lout + BN - 1) / BN;
    const int mtiles = (Cout + BM - 1) / BM;
    const int64_t blocks = B * ptiles * mtiles;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    const size_t lds = Cfg::LDS_FLOATS * sizeof(float);
    auto k = conv_tile_kernel<KS, STRIDE, BM>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_conv1d_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(kThreads), lds, st, x, w, bias, y, Cin, Lin, Cout, Lout, pad, relu, ptiles,
                       mtiles);
    return lbm::check_launch("lbm_conv1d_f32");
}

}  // namespace

extern "C" int lbm_conv1d_f32(const float* x, const float* w, const float* bias, float* y, int64_t B, int32_t Cin,
                              int32_t Lin, int32_t Cout, int32_t ksize, int32_t stride, int32_t pad, int32_t relu,
                              lbm_stream_t stream) {
    LBM_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && Lin >= 1, LBM_EINVAL, "lbm_conv1d_f32: bad sizes");
    LBM_REQUIRE(ksize >= 1 && stride >= 1 && pad >= 0, LBM_EINVAL, "lbm_conv1d_f32: bad ksize/stride/pad");
    LBM_REQUIRE(Lin + 2 * pad >= ksize, LBM_EINVAL, "lbm_conv1d_f32: kernel (%d) larger than padded input (%d)", ksize, Lin + 2 * pad);
    if (B == 0) return LBM_OK;
    LBM_REQUIRE(x && w && y, LBM_EINVAL, "lbm_conv1d_f32: NULL pointer");
    const int Lout = (Lin + 2 * pad - ksize) / stride + 1;
    hipStream_t st = lbm::as_stream(stream);
    const bool w_vec = (((uintptr_t)w & 15) == 0) && ((Cin * ksize) % 4 == 0);
    if (w_vec) {
        const int ck = ksize == 1 ? 32 : 8;   // ConvCfg::CK: a ragged last chunk is shifted, which needs Cin >= CK
        if (Cin < ck) goto generic;
        if (ksize == 4 && stride == 2 && Cout <= 64) return launch_tile<4, 2, 64>(x, w, bias, y, B, Cin, Lin, Cout, Lout, pad, relu, st);
        if (ksize == 4 && stride == 2) return launch_tile<4, 2, 128>(x, w, bias, y, B, Cin, Lin, Cout, Lout, pad, relu, st);
        if (ksize == 3 && stride == 1 && Cin % 4 == 0) return launch_tile<3, 1, 128>(x, w, bias, y, B, Cin, Lin, Cout, Lout, pad, relu, st);
        if (ksize == 1 && stride == 1) return launch_tile<1, 1, 128>(x, w, bias, y, B, Cin, Lin, Cout, Lout, pad, relu, st);
    }
generic:
    const int64_t Ntot = B * Lout;
    const int64_t gx = (Ntot + BN - 1) / BN;
    const int gy = (Cout + GBM - 1) / GBM;
    LBM_REQUIRE(gx <= 0x7fffffffLL && gy <= 65535, LBM_EUNSUPPORTED, "lbm_conv1d_f32: grid too large");
    hipLaunchKernelGGL(conv1d_generic_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(kThreads), 0, st, x, w, bias, y, Ntot, Cin,
                       Lin, Cout, Lout, ksize, stride, pad, relu);
    return lbm::check_launch("lbm_conv1d_f32");
}
