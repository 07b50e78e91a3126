// Epoch post-processing (a4/a5/a6): baseline correction, peak-to-peak rejection mask, z-score.
//
// HBM-bound, one read + up to two writes per sample.  TPR threads cooperate on one (epoch, channel)
// row and keep it in registers (NV values per thread), so the row is read from HBM exactly once:
// TPR = 64 (one wavefront per row, shuffle reductions only, no barrier) up to T = 2560, a 256- or
// 1024-thread workgroup per row beyond that.  Two-pass mean/variance in fp32 on register data.
#include <mutex>

#include "common.h"

namespace {

constexpr int kNV = 40;

template <int TPR>
__device__ __forceinline__ float row_reduce(float v, int op, float* s_red) {
    // op: 0 sum, 1 min, 2 max.  TPR == 64: wavefront only.  Otherwise all waves of the block = one row.
    v = (op == 0) ? lbm::wave_sum(v) : (op == 1) ? lbm::wave_min(v) : lbm::wave_max(v);
    if constexpr (TPR > 64) {
        constexpr int NW = TPR / 64;
        const int w = threadIdx.x >> 6;
        __syncthreads();  // s_red free
        if ((threadIdx.x & 63) == 0) s_red[w] = v;
        __syncthreads();
        float r = s_red[0];
#pragma unroll
        for (int i = 1; i < NW; ++i) r = (op == 0) ? r + s_red[i] : (op == 1) ? fminf(r, s_red[i]) : fmaxf(r, s_red[i]);
        v = r;
    }
    return v;
}

// VEC = 4: rows are 16-byte aligned and T % 4 == 0, every lane moves float4 (16 B/lane streaming);
// VEC = 1: any T / alignment.  Loads are unconditional at a clamped index and masked afterwards (a
// load inside `if (t < T)` makes the compiler branch and drain vmcnt per element).
template <int TPR, int NV, int VEC>
__global__ __launch_bounds__(TPR < 256 ? 256 : TPR) void post_kernel(const float* __restrict__ x, float* y_base,
                                                                      float* y_std, uint8_t* __restrict__ keep,
                                                                      int64_t rows, int C, int T, int nb,
                                                                      float thresh, int std_mode, float eps) {
    __shared__ float s_red[16];
    constexpr int BLOCK = TPR < 256 ? 256 : TPR;
    constexpr int RPB = BLOCK / TPR;
    constexpr int NI = NV / VEC;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
    if (row >= rows) return;  // uniform per row group (wave when TPR == 64, block otherwise)
    const int l = threadIdx.x % TPR;
    const float* xr = x + row * (int64_t)T;

    float v[NV];
    float bsum = 0.f, tsum = 0.f, vmin = INFINITY, vmax = -INFINITY;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t0 = (i * TPR + l) * VEC;
        const int tc = t0 < T ? t0 : 0;   // VEC == 4: T % 4 == 0, so a quad is in range or out as a whole
        if constexpr (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(xr + tc);
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        } else {
            v[i] = xr[tc];
        }
    }
    // all loads are in flight before the first use; the empty asm pins each loaded value so the compiler
    // cannot sink a load back under its `t0 < T` select
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("" : "+v"(v[i]));
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t0 = (i * TPR + l) * VEC;
        const bool ok = t0 < T;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float val = ok ? v[VEC * i + e] : 0.f;
            v[VEC * i + e] = val;
            tsum += val;
            bsum += (t0 + e < nb) ? val : 0.f;
            vmin = fminf(vmin, ok ? val : INFINITY);
            vmax = fmaxf(vmax, ok ? val : -INFINITY);
        }
    }
    tsum = row_reduce<TPR>(tsum, 0, s_red);
    vmin = row_reduce<TPR>(vmin, 1, s_red);
    vmax = row_reduce<TPR>(vmax, 2, s_red);
    float base = 0.f;
    if (nb > 0) base = row_reduce<TPR>(bsum, 0, s_red) / (float)nb;

    if (keep != nullptr && l == 0 && (vmax - vmin) > thresh) keep[row / C] = 0;

    const float mean = tsum / (float)T - base;  // mean of the baseline-corrected row
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t0 = (i * TPR + l) * VEC;
        const bool ok = t0 < T;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[VEC * i + e] -= base;
            const float d = ok ? v[VEC * i + e] - mean : 0.f;
            ss = fmaf(d, d, ss);
        }
        if (ok && y_base != nullptr) {
            if constexpr (VEC == 4)
                *reinterpret_cast<float4*>(y_base + row * (int64_t)T + t0) =
                    make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
            else
                y_base[row * (int64_t)T + t0] = v[i];
        }
    }
    if (y_std == nullptr || std_mode != LBM_STD_EPOCH_CHANNEL) return;
    ss = row_reduce<TPR>(ss, 0, s_red);
    const float denom = sqrtf(ss / (float)T) + eps;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int t0 = (i * TPR + l) * VEC;
        if (t0 < T) {
            if constexpr (VEC == 4)
                *reinterpret_cast<float4*>(y_std + row * (int64_t)T + t0) =
                    make_float4((v[4 * i] - mean) / denom, (v[4 * i + 1] - mean) / denom, (v[4 * i + 2] - mean) / denom,
                                (v[4 * i + 3] - mean) / denom);
            else
                y_std[row * (int64_t)T + t0] = (v[i] - mean) / denom;
        }
    }
}

// one wavefront per (epoch, channel) row: float64 sum and sum of squares
__global__ __launch_bounds__(256) void row_moments_kernel(const float* __restrict__ x, const uint8_t* __restrict__ keep,
                                                          double* __restrict__ row_ws, int64_t rows, int C, int T) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int l = threadIdx.x & 63;
    double s = 0.0, q = 0.0;
    if (keep == nullptr || keep[row / C]) {
        const float* xr = x + row * (int64_t)T;
        for (int t = l; t < T; t += 64) {
            const double v = (double)xr[t];
            s += v;
            q = fma(v, v, q);
        }
    }
    s = lbm::wave_sum(s);
    q = lbm::wave_sum(q);
    if (l == 0) {
        row_ws[2 * row] = s;
        row_ws[2 * row + 1] = q;
    }
}

// one workgroup per channel: fixed-order reduction over epochs -> deterministic moments
__global__ __launch_bounds__(256) void channel_moments_kernel(const double* __restrict__ row_ws,
                                                              const uint8_t* __restrict__ keep,
                                                              double* __restrict__ moments, int64_t E, int C, int T) {
    __shared__ double s_s[4], s_q[4], s_n[4];
    const int c = blockIdx.x;
    double s = 0.0, q = 0.0, n = 0.0;
    for (int64_t e = threadIdx.x; e < E; e += 256) {
        if (keep == nullptr || keep[e]) {
            s += row_ws[2 * (e * C + c)];
            q += row_ws[2 * (e * C + c) + 1];
            n += (double)T;
        }
    }
    s = lbm::wave_sum(s);
    q = lbm::wave_sum(q);
    n = lbm::wave_sum(n);
    if ((threadIdx.x & 63) == 0) {
        s_s[threadIdx.x >> 6] = s;
        s_q[threadIdx.x >> 6] = q;
        s_n[threadIdx.x >> 6] = n;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        moments[3 * c + 0] = (s_s[0] + s_s[1]) + (s_s[2] + s_s[3]);
        moments[3 * c + 1] = (s_q[0] + s_q[1]) + (s_q[2] + s_q[3]);
        moments[3 * c + 2] = (s_n[0] + s_n[1]) + (s_n[2] + s_n[3]);
    }
}

__global__ __launch_bounds__(256) void channel_scale_kernel(const float* x, float* y, const float* __restrict__ mean,
                                                            const float* __restrict__ scale, int64_t rows, int C, int T) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c = (int)(row % C);
    const float m = mean[c], s = scale[c];
    for (int t = threadIdx.x & 63; t < T; t += 64) y[row * (int64_t)T + t] = (x[row * (int64_t)T + t] - m) / s;
}

template <int TPR>
int launch_post(const float* x, float* y_base, float* y_std, uint8_t* keep, int64_t rows, int C, int T, int nb,
                float thresh, int std_mode, float eps, hipStream_t st) {
    constexpr int BLOCK = TPR < 256 ? 256 : TPR;
    constexpr int RPB = BLOCK / TPR;
    const int64_t blocks = (rows + RPB - 1) / RPB;
    if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "lbm_epoch_post_f32: grid too large");
    const bool vec = (T % 4 == 0) && (((uintptr_t)x | (uintptr_t)y_base | (uintptr_t)y_std) & 15) == 0;
    if (vec)
        hipLaunchKernelGGL((post_kernel<TPR, kNV, 4>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, x, y_base, y_std, keep,
                           rows, C, T, nb, thresh, std_mode, eps);
    else
        hipLaunchKernelGGL((post_kernel<TPR, kNV, 1>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, x, y_base, y_std, keep,
                           rows, C, T, nb, thresh, std_mode, eps);
    return lbm::check_launch("lbm_epoch_post_f32");
}

}  // namespace

extern "C" int lbm_epoch_post_f32(const float* x, float* y_base, float* y_std, uint8_t* keep, int64_t E, int32_t C,
                                  int32_t T, int32_t n_baseline, float ptp_thresh, int32_t standardize_mode, float eps,
                                  lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && T >= 0, LBM_EINVAL, "lbm_epoch_post_f32: bad sizes (E=%lld C=%d T=%d)", (long long)E, C, T);
    LBM_REQUIRE(n_baseline >= 0 && n_baseline <= T, LBM_EINVAL, "lbm_epoch_post_f32: n_baseline=%d outside [0,%d]", n_baseline, T);
    LBM_REQUIRE(standardize_mode == LBM_STD_NONE || standardize_mode == LBM_STD_EPOCH_CHANNEL, LBM_EINVAL,
                "lbm_epoch_post_f32: unknown standardize_mode %d", standardize_mode);
    if (E == 0) return LBM_OK;
    hipStream_t st = lbm::as_stream(stream);
    if (keep != nullptr) {
        hipError_t e = hipMemsetAsync(keep, 1, (size_t)E, st);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_epoch_post_f32: memset keep: %s", hipGetErrorString(e));
    }
    if (T == 0) return LBM_OK;
    LBM_REQUIRE(x != nullptr, LBM_EINVAL, "lbm_epoch_post_f32: NULL x");
    LBM_REQUIRE(T <= 1024 * kNV, LBM_EUNSUPPORTED, "lbm_epoch_post_f32: T=%d > %d", T, 1024 * kNV);
    const int64_t rows = E * C;
    if (T <= 64 * kNV) return launch_post<64>(x, y_base, y_std, keep, rows, C, T, n_baseline, ptp_thresh, standardize_mode, eps, st);
    if (T <= 256 * kNV) return launch_post<256>(x, y_base, y_std, keep, rows, C, T, n_baseline, ptp_thresh, standardize_mode, eps, st);
    return launch_post<1024>(x, y_base, y_std, keep, rows, C, T, n_baseline, ptp_thresh, standardize_mode, eps, st);
}

namespace {
int launch_channel_moments(const char* who, const float* x, const uint8_t* keep, double* row_ws, double* moments, int64_t E, int C,
                           int T, hipStream_t st) {
    const int64_t rows = E * C;
    if (rows > 0) {
        const int64_t blocks = (rows + 3) / 4;
        if (blocks > 0x7fffffffLL) return lbm::fail(LBM_EUNSUPPORTED, "%s: grid too large", who);
        hipLaunchKernelGGL(row_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, keep, row_ws, rows, C, T);
    }
    hipLaunchKernelGGL(channel_moments_kernel, dim3(C), dim3(256), 0, st, row_ws, keep, moments, E, C, T);
    return lbm::check_launch(who);
}
}  // namespace

extern "C" int lbm_channel_moments_f32(const float* x, const uint8_t* keep, double* row_ws, double* moments, int64_t E,
                                       int32_t C, int32_t T, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && T >= 0, LBM_EINVAL, "lbm_channel_moments_f32: bad sizes");
    LBM_REQUIRE(moments != nullptr, LBM_EINVAL, "lbm_channel_moments_f32: NULL moments");
    LBM_REQUIRE(E == 0 || (x != nullptr && row_ws != nullptr), LBM_EINVAL, "lbm_channel_moments_f32: NULL pointer");
    return launch_channel_moments("lbm_channel_moments_f32", x, keep, row_ws, moments, E, C, T, lbm::as_stream(stream));
}

// ---- the one entry point that allocates nothing and needs no workspace argument ------------------------------------------
// lbm_channel_moments_f32's per-row partial sums (E*C*2 doubles) live in a file-scope device array instead of a caller buffer.
// Every GPU the library runs on holds its own copy (16 MiB) at its own address until the process ends; the array's NAME is a
// host-side stand-in, so the device address is looked up once per device with hipGetSymbolAddress and only that pointer is
// ever handed to kernels.
static __device__ __align__(256) double scratch[2 * LBM_STATIC_MOMENTS_MAX_ROWS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
double* g_scratch_ptr[kMaxDevices] = {};

// device address of `scratch` on the CURRENT device (looked up at the first call on that device, then cached)
int scratch_of_current_device(double** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_channel_moments_static_f32: hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return lbm::fail(LBM_EUNSUPPORTED, "lbm_channel_moments_static_f32: device %d (max %d)", dev, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr)
            return lbm::fail(LBM_ELAUNCH, "lbm_channel_moments_static_f32: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                             hipGetErrorString(e));
        g_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_scratch_ptr[dev];
    return LBM_OK;
}
}  // namespace

extern "C" int lbm_channel_moments_static_f32(const float* x, const uint8_t* keep, double* moments, int64_t E, int32_t C,
                                              int32_t T, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && T >= 0, LBM_EINVAL, "lbm_channel_moments_static_f32: bad sizes");
    LBM_REQUIRE(moments != nullptr, LBM_EINVAL, "lbm_channel_moments_static_f32: NULL moments");
    LBM_REQUIRE(E == 0 || x != nullptr, LBM_EINVAL, "lbm_channel_moments_static_f32: NULL pointer");
    LBM_REQUIRE(E <= LBM_STATIC_MOMENTS_MAX_ROWS / C, LBM_EUNSUPPORTED,
                "lbm_channel_moments_static_f32: E*C = %lld rows exceed the static scratch (%lld rows): use lbm_channel_moments_f32 "
                "with a caller workspace", (long long)E * C, (long long)LBM_STATIC_MOMENTS_MAX_ROWS);
    double* ws = nullptr;
    const int rc = scratch_of_current_device(&ws);
    if (rc != LBM_OK) return rc;
    // rows [0, E*C) of the scratch are fully rewritten by the row pass before the channel pass reads them: nothing to clear
    return launch_channel_moments("lbm_channel_moments_static_f32", x, keep, ws, moments, E, C, T, lbm::as_stream(stream));
}

extern "C" int lbm_channel_scale_f32(const float* x, float* y, const float* mean, const float* scale, int64_t E,
                                     int32_t C, int32_t T, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && T >= 0, LBM_EINVAL, "lbm_channel_scale_f32: bad sizes");
    if (E == 0 || T == 0) return LBM_OK;
    LBM_REQUIRE(x && y && mean && scale, LBM_EINVAL, "lbm_channel_scale_f32: NULL pointer");
    const int64_t rows = E * C;
    const int64_t blocks = (rows + 3) / 4;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_channel_scale_f32: grid too large");
    hipLaunchKernelGGL(channel_scale_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), x, y, mean, scale,
                       rows, C, T);
    return lbm::check_launch("lbm_channel_scale_f32");
}
