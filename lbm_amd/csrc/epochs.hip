// Epoch slicing (a4, first half): mne.Epochs(raw, events, tmin, tmax) window extraction from a continuous
// recording (training/preprocess.py:262-271, multi-train/withpsd.py:213-216).  Pure data movement:
// out[e, c, t] = raw[c, start[e] + t].  One wavefront per (epoch, channel) row, coalesced 4-byte accesses
// (window starts are arbitrary sample indices, so rows are not 16-byte aligned in general).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void epochs_gather_kernel(const float* __restrict__ raw, int64_t ldr,
                                                            const int64_t* __restrict__ starts, float* __restrict__ out,
                                                            int64_t rows, int C, int Tw, int64_t Trec) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int64_t e = row / C;
    const int c = (int)(row - e * C);
    const int64_t s = starts[e];
    const float* src = raw + (int64_t)c * ldr;
    float* dst = out + row * (int64_t)Tw;
    for (int t = threadIdx.x & 63; t < Tw; t += 64) {
        const int64_t m = s + t;
        dst[t] = (m >= 0 && m < Trec) ? src[m] : 0.f;   // callers drop out-of-range events; stay in bounds anyway
    }
}

}  // namespace

extern "C" int lbm_epochs_gather_f32(const float* raw, int64_t ldr, const int64_t* starts, float* out, int64_t E, int32_t C,
                                     int32_t Tw, int64_t Trec, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && Tw >= 0 && Trec >= 0 && ldr >= Trec, LBM_EINVAL, "lbm_epochs_gather_f32: bad sizes");
    if (E == 0 || Tw == 0) return LBM_OK;
    LBM_REQUIRE(raw && starts && out, LBM_EINVAL, "lbm_epochs_gather_f32: NULL pointer");
    const int64_t rows = E * C;
    const int64_t blocks = (rows + 3) / 4;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_epochs_gather_f32: grid too large");
    hipLaunchKernelGGL(epochs_gather_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), raw, ldr, starts, out,
                       rows, C, Tw, Trec);
    return lbm::check_launch("lbm_epochs_gather_f32");
}
