// Welch PSD (a7) and band powers (a8).
//
// Tables (caller-owned device buffer, filled once by lbm_welch_tables_init in float64):
//   bytes 0..7   double  sum(w^2)
//   bytes 64..   float   window[nperseg4]            (periodic Hann / Hamming / boxcar)
//                float   cos[nfft4], sin[nfft4]      (twiddles of 2*pi*j/nfft)
//
// lbm_welch_f32: one workgroup per row.  The row's segments are mean-removed, windowed and staged in
// LDS; each thread then evaluates the pruned DFT (only bins bin_lo..bin_hi, e.g. 27 of 251) of
// (segment, bin) pairs with an exact integer twiddle index (k*n mod nfft), and the segment average is
// taken in a fixed order.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kHeader = 64;  // bytes

__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }

struct TableView {
    const double* sum_w2;
    const float* window;
    const float* cosv;
    const float* sinv;
};
__host__ __device__ inline TableView view_tables(const void* tables, int nfft, int nperseg) {
    const char* p = reinterpret_cast<const char*>(tables);
    TableView v;
    v.sum_w2 = reinterpret_cast<const double*>(p);
    v.window = reinterpret_cast<const float*>(p + kHeader);
    v.cosv = v.window + pad4(nperseg);
    v.sinv = v.cosv + pad4(nfft);
    return v;
}

__device__ inline double window_value(int window_id, int n, int N) {
    const double c = cospi(2.0 * (double)n / (double)N);
    if (window_id == LBM_WIN_HANN) return 0.5 - 0.5 * c;
    if (window_id == LBM_WIN_HAMMING) return 0.54 - 0.46 * c;
    return 1.0;
}

__global__ __launch_bounds__(kThreads) void welch_tables_kernel(void* tables, int nfft, int nperseg, int window_id) {
    __shared__ double s_part[kThreads];
    char* p = reinterpret_cast<char*>(tables);
    float* window = reinterpret_cast<float*>(p + kHeader);
    float* cosv = window + pad4(nperseg);
    float* sinv = cosv + pad4(nfft);
    double acc = 0.0;
    for (int n = threadIdx.x; n < pad4(nperseg); n += kThreads) {
        const double w = (n < nperseg) ? window_value(window_id, n, nperseg) : 0.0;
        window[n] = (float)w;
        acc += w * w;  // float64 window, as scipy's scale uses
    }
    for (int j = threadIdx.x; j < pad4(nfft); j += kThreads) {
        const double a = 2.0 * (double)j / (double)nfft;
        cosv[j] = (j < nfft) ? (float)cospi(a) : 0.f;
        sinv[j] = (j < nfft) ? (float)sinpi(a) : 0.f;
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < kThreads; ++i) s += s_part[i];
        *reinterpret_cast<double*>(p) = s;
    }
}

__global__ __launch_bounds__(kThreads) void welch_kernel(const float* __restrict__ x, float* __restrict__ psd,
                                                         float* __restrict__ seg_out, int T, double fs, int nfft,
                                                         int nperseg, int hop, int nseg, const void* __restrict__ tables,
                                                         int bin_lo, int F, int average_mode) {
    extern __shared__ __align__(16) float smem[];
    float* s_cos = smem;                          // [nfft]
    float* s_sin = s_cos + pad4(nfft);            // [nfft]
    float* s_xw = s_sin + pad4(nfft);             // [nseg][nperseg]
    float* s_p = s_xw + (size_t)nseg * nperseg;   // [nseg][F]
    const TableView tv = view_tables(tables, nfft, nperseg);
    const int64_t row = blockIdx.x;
    const float* xr = x + row * (int64_t)T;

    for (int j = threadIdx.x; j < nfft; j += kThreads) {
        s_cos[j] = tv.cosv[j];
        s_sin[j] = tv.sinv[j];
    }
    // per-segment mean removal + window: one wavefront per segment
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int s = wave; s < nseg; s += kThreads / 64) {
        const float* xs = xr + (int64_t)s * hop;
        float sum = 0.f;
        for (int n = lane; n < nperseg; n += 64) sum += xs[n];
        const float mean = lbm::wave_sum(sum) / (float)nperseg;
        for (int n = lane; n < nperseg; n += 64) s_xw[(size_t)s * nperseg + n] = (xs[n] - mean) * tv.window[n];
    }
    __syncthreads();

    const double scale = 1.0 / (fs * (*tv.sum_w2));
    const int nout = nseg * F;
    for (int o = threadIdx.x; o < nout; o += kThreads) {
        const int s = o / F, k = bin_lo + (o % F);
        const float* xs = s_xw + (size_t)s * nperseg;
        float re = 0.f, im = 0.f;
        int idx = 0;
        for (int n = 0; n < nperseg; ++n) {
            const float v = xs[n];
            re = fmaf(v, s_cos[idx], re);
            im = fmaf(v, s_sin[idx], im);
            idx += k;
            if (idx >= nfft) idx -= nfft;
        }
        const bool one_sided = (k == 0) || ((nfft & 1) == 0 && k == nfft / 2);
        const double p = ((double)re * re + (double)im * im) * scale * (one_sided ? 1.0 : 2.0);
        s_p[o] = (float)p;
    }
    __syncthreads();

    for (int b = threadIdx.x; b < F; b += kThreads) {
        if (seg_out != nullptr)
            for (int s = 0; s < nseg; ++s) seg_out[(row * F + b) * (int64_t)nseg + s] = s_p[s * F + b];
        if (average_mode == LBM_AVG_MEAN) {
            float acc = 0.f;
            for (int s = 0; s < nseg; ++s) acc += s_p[s * F + b];
            psd[row * F + b] = acc / (float)nseg;
        } else if (average_mode == LBM_AVG_MEDIAN) {
            // rank selection without reordering; ties resolved by position
            float lo = 0.f, hi = 0.f;
            const int r_lo = (nseg - 1) / 2, r_hi = nseg / 2;
            for (int s = 0; s < nseg; ++s) {
                const float v = s_p[s * F + b];
                int rank = 0;
                for (int u = 0; u < nseg; ++u) {
                    const float w = s_p[u * F + b];
                    rank += (w < v) || (w == v && u < s);
                }
                if (rank == r_lo) lo = v;
                if (rank == r_hi) hi = v;
            }
            double bias = 1.0;  // scipy.signal._spectral_py._median_bias
            for (int i = 2; i <= nseg; ++i) bias += ((i & 1) ? 1.0 : -1.0) / (double)i;
            psd[row * F + b] = (float)((0.5 * ((double)lo + (double)hi)) / bias);
        }
    }
}

// one wavefront per epoch (or per (epoch, channel) row when per_channel)
__global__ __launch_bounds__(256) void bandpower_kernel(const float* __restrict__ psd, const float* __restrict__ freqs,
                                                        float* __restrict__ out, int64_t groups, int n_elem, int F,
                                                        int already_db) {
    const int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int lane = threadIdx.x & 63;
    const float* p = psd + g * (int64_t)n_elem;
    double st = 0.0, sa = 0.0, sb = 0.0;  // float64 accumulation of the fp32 dB values
    int nt = 0, na = 0, nb = 0;
    for (int i = lane; i < n_elem; i += 64) {
        const float f = freqs[i % F];
        const double v = already_db ? (double)p[i] : (double)(10.f * log10f(p[i]));
        if (f >= 4.f && f < 8.f) { st += v; ++nt; }
        else if (f >= 8.f && f < 13.f) { sa += v; ++na; }
        else if (f >= 13.f && f < 30.f) { sb += v; ++nb; }
    }
    st = lbm::wave_sum(st); sa = lbm::wave_sum(sa); sb = lbm::wave_sum(sb);
    const double ct = lbm::wave_sum((double)nt), ca = lbm::wave_sum((double)na), cb = lbm::wave_sum((double)nb);
    if (lane == 0) {
        const double theta = st / ct, alpha = sa / ca, beta = sb / cb;
        float* o = out + g * 6;
        o[0] = (float)theta; o[1] = (float)alpha; o[2] = (float)beta;
        o[3] = (float)(beta / theta); o[4] = (float)(alpha / beta); o[5] = (float)(theta / beta);
    }
}

size_t welch_lds_bytes(int nfft, int nperseg, int nseg, int F) {
    return ((size_t)2 * pad4(nfft) + (size_t)nseg * nperseg + (size_t)nseg * F) * sizeof(float);
}

}  // namespace

extern "C" size_t lbm_welch_tables_bytes(int32_t nfft, int32_t nperseg, int32_t bin_lo, int32_t bin_hi) {
    (void)bin_lo; (void)bin_hi;
    if (nfft < 1 || nperseg < 1) return 0;
    return (size_t)kHeader + ((size_t)pad4(nperseg) + 2 * (size_t)pad4(nfft)) * sizeof(float);
}

extern "C" int lbm_welch_tables_init(void* tables, int32_t nfft, int32_t nperseg, int32_t window_id, int32_t bin_lo,
                                     int32_t bin_hi, lbm_stream_t stream) {
    LBM_REQUIRE(tables != nullptr && ((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_welch_tables_init: tables NULL or not 16-byte aligned");
    LBM_REQUIRE(nperseg >= 1 && nperseg <= nfft, LBM_EINVAL, "lbm_welch_tables_init: need 1 <= nperseg <= nfft");
    LBM_REQUIRE(window_id >= LBM_WIN_BOXCAR && window_id <= LBM_WIN_HAMMING, LBM_EINVAL, "lbm_welch_tables_init: unknown window_id %d", window_id);
    LBM_REQUIRE(bin_lo >= 0 && bin_lo <= bin_hi && bin_hi <= nfft / 2, LBM_EINVAL, "lbm_welch_tables_init: bins [%d,%d] outside [0,%d]", bin_lo, bin_hi, nfft / 2);
    hipLaunchKernelGGL(welch_tables_kernel, dim3(1), dim3(kThreads), 0, lbm::as_stream(stream), tables, nfft, nperseg, window_id);
    return lbm::check_launch("lbm_welch_tables_init");
}

extern "C" int lbm_welch_f32(const float* x, float* psd, float* seg_or_null, int64_t rows, int32_t T, float fs,
                             int32_t nfft, int32_t nperseg, int32_t noverlap, const void* tables, int32_t bin_lo,
                             int32_t bin_hi, int32_t average_mode, lbm_stream_t stream) {
    LBM_REQUIRE(rows >= 0, LBM_EINVAL, "lbm_welch_f32: negative rows");
    LBM_REQUIRE(nperseg >= 1 && nperseg <= nfft, LBM_EINVAL, "lbm_welch_f32: need 1 <= nperseg <= nfft");
    LBM_REQUIRE(nfft <= T, LBM_EINVAL, "lbm_welch_f32: n_fft (%d) is larger than the number of samples (%d)", nfft, T);
    LBM_REQUIRE(noverlap >= 0 && noverlap < nperseg, LBM_EINVAL, "lbm_welch_f32: need 0 <= noverlap < nperseg");
    LBM_REQUIRE(bin_lo >= 0 && bin_lo <= bin_hi && bin_hi <= nfft / 2, LBM_EINVAL, "lbm_welch_f32: bins [%d,%d] outside [0,%d]", bin_lo, bin_hi, nfft / 2);
    LBM_REQUIRE(average_mode >= LBM_AVG_MEAN && average_mode <= LBM_AVG_NONE, LBM_EINVAL, "lbm_welch_f32: unknown average_mode %d", average_mode);
    LBM_REQUIRE(fs > 0.f, LBM_EINVAL, "lbm_welch_f32: fs must be positive");
    if (rows == 0) return LBM_OK;
    LBM_REQUIRE(x != nullptr && tables != nullptr, LBM_EINVAL, "lbm_welch_f32: NULL pointer");
    LBM_REQUIRE(average_mode == LBM_AVG_NONE ? seg_or_null != nullptr : psd != nullptr, LBM_EINVAL, "lbm_welch_f32: NULL output");
    LBM_REQUIRE(rows <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_welch_f32: too many rows");
    const int hop = nperseg - noverlap;
    const int nseg = (T - noverlap) / hop;
    const int F = bin_hi - bin_lo + 1;
    const size_t lds = welch_lds_bytes(nfft, nperseg, nseg, F);
    LBM_REQUIRE(lds <= 160 * 1024, LBM_EUNSUPPORTED, "lbm_welch_f32: row needs %zu B of LDS (max 163840)", lds);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(welch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_welch_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(welch_kernel, dim3((unsigned)rows), dim3(kThreads), lds, lbm::as_stream(stream), x, psd, seg_or_null, T,
                       (double)fs, nfft, nperseg, hop, nseg, tables, bin_lo, F, average_mode);
    return lbm::check_launch("lbm_welch_f32");
}

extern "C" int lbm_bandpower_f32(const float* psd, const float* freqs, float* out, int64_t E, int32_t C, int32_t F,
                                 int32_t per_channel, int32_t already_db, lbm_stream_t stream) {
    LBM_REQUIRE(E >= 0 && C >= 1 && F >= 1, LBM_EINVAL, "lbm_bandpower_f32: bad sizes");
    if (E == 0) return LBM_OK;
    LBM_REQUIRE(psd && freqs && out, LBM_EINVAL, "lbm_bandpower_f32: NULL pointer");
    const int64_t groups = per_channel ? E * C : E;
    const int n_elem = per_channel ? F : C * F;
    const int64_t blocks = (groups + 3) / 4;
    LBM_REQUIRE(blocks <= 0x7fffffffLL, LBM_EUNSUPPORTED, "lbm_bandpower_f32: grid too large");
    hipLaunchKernelGGL(bandpower_kernel, dim3((unsigned)blocks), dim3(256), 0, lbm::as_stream(stream), psd, freqs, out, groups,
                       n_elem, F, already_db);
    return lbm::check_launch("lbm_bandpower_f32");
}
