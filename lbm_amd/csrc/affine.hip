// Per-sample channel affine map y[e,o,t] = sum_c A[o,c] x[e,c,t] + b[o]  (a3: ICA apply / unmix).
//
// HBM-bound (2*Cin*Cout flops per Cin+Cout floats moved).  A thread owns VEC consecutive time samples
// of one epoch and up to 32 output channels: it streams the Cin input rows once (coalesced, 16 B per
// lane when T % 4 == 0) and keeps the 32 x VEC accumulators in registers; A^T lives in LDS and is read
// with broadcast ds_read_b128.  Output channels beyond 32 are handled by blockIdx.y (the input rows are
// then re-read through L2).  fp32 FMA chain in ascending input-channel order, bias added last.
#include "common.h"

namespace {

constexpr int kThreads = 128;
constexpr int kCoBlk = 32;

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using type = float4;
};
template <>
struct Vec<1> {
    using type = float;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
        v[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        *p = v[0];
    }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void affine_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int Cin, int Cout, int T, const float* __restrict__ A,
                                                          const float* __restrict__ b, int tchunks) {
    extern __shared__ __align__(16) float smem[];
    float* s_at = smem;                 // [Cin][kCoBlk]  A^T slice, zero padded
    float* s_b = smem + Cin * kCoBlk;   // [kCoBlk]
    const int co0 = blockIdx.y * kCoBlk;
    for (int i = threadIdx.x; i < Cin * kCoBlk; i += kThreads) {
        const int ci = i / kCoBlk, co = co0 + (i % kCoBlk);
        s_at[i] = (co < Cout) ? A[(int64_t)co * Cin + ci] : 0.f;
    }
    if (threadIdx.x < kCoBlk) {
        const int co = co0 + threadIdx.x;
        s_b[threadIdx.x] = (b != nullptr && co < Cout) ? b[co] : 0.f;
    }
    __syncthreads();

    const int64_t e = blockIdx.x / tchunks;
    const int t = ((blockIdx.x % tchunks) * kThreads + threadIdx.x) * VEC;
    if (t >= T) return;
    const float* xp = x + e * (int64_t)Cin * T + t;
    float acc[kCoBlk][VEC];
#pragma unroll
    for (int o = 0; o < kCoBlk; ++o)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[o][v] = 0.f;

#pragma unroll 2
    for (int ci = 0; ci < Cin; ++ci) {
        float xv[VEC];
        load_vec<VEC>(xp + (int64_t)ci * T, xv);
        const float4* at = reinterpret_cast<const float4*>(s_at + ci * kCoBlk);
#pragma unroll
        for (int q = 0; q < kCoBlk / 4; ++q) {
            const float4 a4 = at[q];
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                acc[4 * q + 0][v] = fmaf(a4.x, xv[v], acc[4 * q + 0][v]);
                acc[4 * q + 1][v] = fmaf(a4.y, xv[v], acc[4 * q + 1][v]);
                acc[4 * q + 2][v] = fmaf(a4.z, xv[v], acc[4 * q + 2][v]);
                acc[4 * q + 3][v] = fmaf(a4.w, xv[v], acc[4 * q + 3][v]);
            }
        }
    }
    float* yp = y + e * (int64_t)Cout * T + t;
#pragma unroll
    for (int o = 0; o < kCoBlk; ++o) {
        if (co0 + o < Cout) {
            float out[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) out[v] = acc[o][v] + s_b[o];
            store_vec<VEC>(yp + (int64_t)(co0 + o) * T, out);
        }
    }
}

}  // namespace

extern "C" int lbm_affine_ch_f32(const float* x, float* y, int64_t E, int32_t Cin, int32_t Cout, int32_t T,
                                 const float* A, const float* b, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && T >= 0, LBM_EINVAL, "lbm_affine_ch_f32: negative size");
    LBM_REQUIRE(Cin >= 1 && Cin <= 256 && Cout >= 1 && Cout <= 256, LBM_EINVAL,
                "lbm_affine_ch_f32: channel counts must be in [1,256] (Cin=%d, Cout=%d)", Cin, Cout);
    if (E == 0 || T == 0) return LBM_OK;
    LBM_REQUIRE(x && y && A, LBM_EINVAL, "lbm_affine_ch_f32: NULL pointer");
    const bool vec4 = (T % 4 == 0) && (((uintptr_t)x | (uintptr_t)y) & 15) == 0;
    const int vec = vec4 ? 4 : 1;
    const int tchunks = (T + kThreads * vec - 1) / (kThreads * vec);
    const int64_t bx = E * tchunks;
    LBM_REQUIRE(bx <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_affine_ch_f32: grid too large");
    const dim3 grid((unsigned)bx, (unsigned)((Cout + kCoBlk - 1) / kCoBlk));
    const size_t lds = (size_t)(Cin * kCoBlk + kCoBlk) * sizeof(float);
    if (vec4)
        hipLaunchKernelGGL(affine_kernel<4>, grid, dim3(kThreads), lds, lbm::as_stream(stream), x, y, Cin, Cout, T, A, b,
                           tchunks);
    else
        hipLaunchKernelGGL(affine_kernel<1>, grid, dim3(kThreads), lds, lbm::as_stream(stream), x, y, Cin, Cout, T, A, b,
                           tchunks);
    return lbm::check_launch("lbm_affine_ch_f32");
}
