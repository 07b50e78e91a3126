// Shared host-side helpers for liblbm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/lbm_abi.h"

namespace lbm {

// thread-local last-error message, written by every failing entry point
char* last_error_buf();
int fail(int code, const char* fmt, ...);

inline hipStream_t as_stream(lbm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// after a kernel launch: report (and clear) a launch error without synchronising
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LBM_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return LBM_OK;
}

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace lbm

#define LBM_REQUIRE(cond, code, ...) \
    do {                             \
        if (!(cond)) return lbm::fail(code, __VA_ARGS__); \
    } while (0)
