// Zero-phase FIR over [rows, T] (a1/a2): LDS-staged, register-blocked direct form.
//
// One workgroup = one (row, tile of 1536 outputs).  The padded input span of the tile
// (tile + ntaps - 1 samples, odd-reflected at the row ends) and the reversed taps are staged in LDS
// once; every thread then produces 8 consecutive outputs from a 12-sample sliding register window,
// so 4 taps x 8 outputs = 32 FMAs are fed by two ds_read_b128 (one broadcast read of the taps, one
// of the window's 4 new samples).  fp32 FMA chain in ascending tap order.
#include "common.h"

namespace {

constexpr int kThreads = 192;
constexpr int kPerThread = 8;
constexpr int kTile = kThreads * kPerThread;  // 1536 outputs

__device__ __forceinline__ float padded_sample(const float* __restrict__ xr, int m, int T, int pad_mode) {
    if (m >= 0 && m < T) return xr[m];
    if (pad_mode != LBM_PAD_REFLECT_LIMITED) return 0.f;
    if (m < 0) {
        int mm = -m;
        return (mm <= T - 1) ? 2.f * xr[0] - xr[mm] : 0.f;
    }
    int mm = 2 * (T - 1) - m;
    return (mm >= 0) ? 2.f * xr[T - 1] - xr[mm] : 0.f;
}

#define LBM_FIR_STEP(A, B, C)                                                   \
    {                                                                           \
        C = *reinterpret_cast<const float4*>(s_x + n0 + j + 8);                 \
        const float4 tp = *reinterpret_cast<const float4*>(s_taps + j);         \
        const float w[12] = {A.x, A.y, A.z, A.w, B.x, B.y, B.z, B.w, C.x, C.y, C.z, C.w}; \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) acc[r] = fmaf(tp.x, w[r], acc[r]);     \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) acc[r] = fmaf(tp.y, w[r + 1], acc[r]); \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) acc[r] = fmaf(tp.z, w[r + 2], acc[r]); \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) acc[r] = fmaf(tp.w, w[r + 3], acc[r]); \
        j += 4;                                                                 \
    }

__global__ __launch_bounds__(kThreads) void fir_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       int T, const float* __restrict__ taps, int ntaps,
                                                       int ntaps4, int pad_mode, int tiles_per_row) {
    extern __shared__ __align__(16) float smem[];
    float* s_taps = smem;          // [ntaps4] reversed taps, zero padded
    float* s_x = smem + ntaps4;    // [kTile + ntaps4] padded input span of this tile

    const int64_t row = blockIdx.x / tiles_per_row;
    const int tile = blockIdx.x % tiles_per_row;
    const int n_base = tile * kTile;
    const int half = (ntaps - 1) >> 1;
    const float* xr = x + row * (int64_t)T;

    for (int i = threadIdx.x; i < ntaps4; i += kThreads) s_taps[i] = (i < ntaps) ? taps[ntaps - 1 - i] : 0.f;
    const int span = kTile + ntaps4;
    for (int i = threadIdx.x; i < span; i += kThreads)
        s_x[i] = padded_sample(xr, n_base + i - half, T, pad_mode);
    __syncthreads();

    const int n0 = threadIdx.x * kPerThread;
    if (n_base + n0 >= T) return;  // no barrier after this point

    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    float4 a = *reinterpret_cast<const float4*>(s_x + n0);
    float4 b = *reinterpret_cast<const float4*>(s_x + n0 + 4);
    float4 c;
    int j = 0;
    const int n3 = ntaps4 - (ntaps4 % 12);
    while (j < n3) {
        LBM_FIR_STEP(a, b, c)
        LBM_FIR_STEP(b, c, a)
        LBM_FIR_STEP(c, a, b)
    }
    while (j < ntaps4) {
        LBM_FIR_STEP(a, b, c)
        a = b;
        b = c;
    }

    float* yr = y + row * (int64_t)T + n_base + n0;
    const int left = T - (n_base + n0);
    if (left >= 8 && (T & 3) == 0) {
        *reinterpret_cast<float4*>(yr) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(yr + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < left) yr[r] = acc[r];
    }
}

}  // namespace

extern "C" int lbm_fir_f32(const float* x, float* y, int64_t rows, int32_t T, const float* taps, int32_t ntaps,
                           int32_t pad_mode, lbm_stream_t stream) {
    LBM_REQUIRE(rows >= 0 && T >= 0, LBM_EINVAL, "lbm_fir_f32: negative size (rows=%lld, T=%d)", (long long)rows, T);
    if (rows == 0 || T == 0) return LBM_OK;
    LBM_REQUIRE(x && y && taps, LBM_EINVAL, "lbm_fir_f32: NULL pointer");
    LBM_REQUIRE(ntaps >= 1 && (ntaps & 1), LBM_EINVAL, "lbm_fir_f32: ntaps must be odd and >= 1 (got %d)", ntaps);
    LBM_REQUIRE(pad_mode == LBM_PAD_REFLECT_LIMITED || pad_mode == LBM_PAD_ZERO, LBM_EINVAL,
                "lbm_fir_f32: unknown pad_mode %d", pad_mode);
    LBM_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 15) == 0, LBM_EINVAL,
                "lbm_fir_f32: x must be 4-byte and y 16-byte aligned");
    const int ntaps4 = (ntaps + 3) & ~3;
    const size_t lds = (size_t)(2 * ntaps4 + kTile) * sizeof(float);
    LBM_REQUIRE(lds <= 160 * 1024, LBM_EUNSUPPORTED, "lbm_fir_f32: ntaps=%d needs %zu B of LDS (max 163840)", ntaps, lds);
    const int tiles_per_row = (T + kTile - 1) / kTile;
    const int64_t blocks = rows * tiles_per_row;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_fir_f32: too many tiles (%lld)", (long long)blocks);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fir_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_fir_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(fir_kernel, dim3((unsigned)blocks), dim3(kThreads), lds, lbm::as_stream(stream), x, y, T, taps,
                       ntaps, ntaps4, pad_mode, tiles_per_row);
    return lbm::check_launch("lbm_fir_f32");
}
