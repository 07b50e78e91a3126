// FFT resampling of a continuous recording: raw.resample(250) at
// preprocessing_unsupervised.py:37 (MNE `resample(x, up, down, npad='auto', window='boxcar', pad='reflect_limited')`):
//   pad the row to N = the next power of two (odd reflection, zeros where the reflection runs out), FFT,
//   keep the new_len/2 + 1 lowest bins (Nyquist bin doubled when shortening / halved when lengthening),
//   inverse FFT at new_len = ratio * N, scale by ratio, drop the resampled padding.
// The transforms are one or two million points per channel, far beyond a workgroup: Stockham autosort passes
// over HBM (radix 16, one thread = one butterfly, natural order in and out, ping-pong between two complex
// buffers), twiddles from sincospi in float64 rounded once.  HBM-bound: (passes + 2) x rows x N x 8 B each way.
// Supported: ratios 2^s (250 <- 500, 250 <- 1000, 500 <- 250 ...), which keep new_len a power of two.
#include "common.h"
#include "fft_small.h"

namespace {

struct Plan {
    int64_t T, N, Nn, final_len, pad0, rem0;   // input, padded, resampled padded, output length; left pad; left trim
    int64_t M;                                 // 0: Nn is a power of two (plain inverse FFT); else the chirp-z convolution length
    int ok;
};

Plan make_plan(int64_t T, int up, int down) {
    Plan p{};
    p.T = T;
    if (T < 1 || up < 1 || down < 1) return p;
    // (llrint: round-half-even, as Python's round() in MNE's length arithmetic)
    // MNE npad='auto' (mne/filter.py resample): min_add = min(x_len // 8, 100) * 2; npad = 2^ceil(log2(x_len + min_add)) - x_len
    // (the doubling is already inside min_add; round 1 doubled it again -- ADVICE r1)
    const int64_t min_add = (T / 8 < 100 ? T / 8 : 100) * 2;
    int64_t N = 1;
    while (N < T + min_add) N <<= 1;
    const int64_t npad = N - T;
    p.N = N;
    p.pad0 = npad / 2;
    const double ratio = (double)up / (double)down;
    p.final_len = (int64_t)llrint(ratio * (double)T);
    p.Nn = (int64_t)llrint(ratio * (double)N);
    if (p.Nn < 1) p.Nn = 1;
    p.rem0 = (int64_t)llrint(ratio * (double)p.pad0);
    const bool pow2 = (p.Nn & (p.Nn - 1)) == 0 && p.Nn >= 16;
    p.M = 0;
    if (!pow2) {   // any other length: the inverse transform of length Nn as a chirp-z (Bluestein) convolution of length M = 2^m >= 2 Nn - 1
        int64_t M = 16;
        while (M < 2 * p.Nn - 1) M <<= 1;
        p.M = M;
    }
    p.ok = N >= 16 && p.final_len >= 1 && N <= (1LL << 30) && p.Nn <= (1LL << 29) && p.rem0 + p.final_len <= p.Nn;
    return p;
}

// row -> complex padded row (im = 0): xp[i] = x[i - pad0] inside, odd reflection 2 x[0] - x[m] / 2 x[T-1] - x[T-1-m]
// outside while m <= T - 1, zero beyond (MNE _smart_pad 'reflect_limited')
__global__ __launch_bounds__(256) void resample_pad_kernel(const float* __restrict__ x, int64_t ldx, float2* __restrict__ a,
                                                           int64_t T, int64_t N, int64_t pad0) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float* xr = x + (int64_t)blockIdx.y * ldx;
    const int64_t s = i - pad0;
    float v;
    if (s >= 0 && s < T) v = xr[s];
    else if (s < 0) v = (-s <= T - 1) ? 2.f * xr[0] - xr[-s] : 0.f;
    else {
        const int64_t m = s - (T - 1);
        v = (m <= T - 1) ? 2.f * xr[T - 1] - xr[T - 1 - m] : 0.f;
    }
    a[(int64_t)blockIdx.y * N + i] = make_float2(v, 0.f);
}

// One Stockham pass of radix R over rows of N complex points: Ns = product of the radices already applied.
// INV: conjugate twiddles and butterflies (unnormalised inverse).
template <int R, bool INV>
__global__ __launch_bounds__(256) void stockham_kernel(const float2* __restrict__ in, float2* __restrict__ out, int64_t N, int64_t Ns) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nb = N / R;
    if (j >= nb) return;
    const float2* src = in + (int64_t)blockIdx.y * N;
    float2* dst = out + (int64_t)blockIdx.y * N;
    const int64_t k = j % Ns;
    float re[R], im[R];
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const float2 v = src[j + t * nb];
        re[t] = v.x;
        im[t] = v.y;
    }
    if (Ns > 1) {
        const double base = -2.0 * (double)k / (double)(Ns * R);   // angle / pi of exp(-2 pi i k t / (Ns R)) per unit t
#pragma unroll
        for (int t = 1; t < R; ++t) {
            double sn, cs;
            sincospi(base * t, &sn, &cs);
            const float c = (float)cs, s = INV ? -(float)sn : (float)sn;
            const float r = re[t] * c - im[t] * s, q = re[t] * s + im[t] * c;
            re[t] = r;
            im[t] = q;
        }
    }
    if constexpr (R == 2) {
        const float ar = re[0] + re[1], ai = im[0] + im[1], br = re[0] - re[1], bi = im[0] - im[1];
        re[0] = ar; im[0] = ai; re[1] = br; im[1] = bi;
    } else {
        if (INV) lbm::dft<R>(im, re);   // idft(re, im) == dft(im, re)
        else lbm::dft<R>(re, im);
    }
    const int64_t j0 = (j / Ns) * Ns * R + k;
#pragma unroll
    for (int t = 0; t < R; ++t) dst[j0 + t * Ns] = make_float2(re[t], im[t]);
}

// Hermitian spectrum of the resampled row from the spectrum of the padded row.  shorter: Nn < N.
__global__ __launch_bounds__(256) void resample_spectrum_kernel(const float2* __restrict__ X, float2* __restrict__ Z, int64_t N,
                                                                int64_t Nn) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= Nn) return;
    const float2* xr = X + (int64_t)blockIdx.y * N;
    const bool shorter = Nn < N;
    const int64_t use = shorter ? Nn : N;          // rfft bins 0 .. use/2 carry over
    const int64_t half = use / 2;
    const int64_t kk = k <= Nn / 2 ? k : Nn - k;   // bin of the half spectrum this entry mirrors
    float2 v = make_float2(0.f, 0.f);
    if (kk <= half) {
        v = xr[kk];
        if (kk == half && (use % 2 == 0)) {        // x_fft[nyq] *= 2 if shorter else 0.5
            const float f = shorter ? 2.f : 0.5f;
            v.x *= f;
            v.y *= f;
        }
        if (kk == 0 || (kk == Nn / 2 && Nn % 2 == 0)) v.y = 0.f;   // irfft ignores the imaginary part of DC / Nyquist
        if (k > Nn / 2) v.y = -v.y;
    }
    Z[(int64_t)blockIdx.y * Nn + k] = v;
}

__global__ __launch_bounds__(256) void resample_trim_kernel(const float2* __restrict__ z, float* __restrict__ y, int64_t ldy, int64_t Nn,
                                                            int64_t rem0, int64_t final_len, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= final_len) return;
    y[(int64_t)blockIdx.y * ldy + i] = z[(int64_t)blockIdx.y * Nn + rem0 + i].x * scale;
}

// ---- inverse transform of arbitrary length L = Nn as a chirp-z convolution (Bluestein):
//   y[n] = sum_k Z[k] e^{+2 pi i n k / L} = c[n] * sum_k (Z[k] c[k]) conj(c[n - k]),   c[n] = e^{i pi n^2 / L}
// chirp[n] for n < L (phase from n^2 mod 2L in exact integer arithmetic, evaluated in float64, rounded once) and the wrapped
// kernel b[m] = conj(c[m]) at m and M - m, zero elsewhere
__global__ __launch_bounds__(256) void chirp_kernel(float2* __restrict__ chirp, float2* __restrict__ b, int64_t L, int64_t M) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float2 bv = make_float2(0.f, 0.f);
    const int64_t n = m < L ? m : (M - m < L ? M - m : -1);
    if (n >= 0) {
        const int64_t q = (int64_t)(((unsigned long long)n * (unsigned long long)n) % (unsigned long long)(2 * L));   // n < 2^29: n^2 < 2^58
        double sn, cs;
        sincospi((double)q / (double)L, &sn, &cs);
        bv = make_float2((float)cs, -(float)sn);
        if (m < L) chirp[m] = make_float2((float)cs, (float)sn);
    }
    b[m] = bv;
}

// A[k] = Z[k] c[k] for k < L (Z = the Hermitian spectrum resample_spectrum_kernel builds), 0 up to M
__global__ __launch_bounds__(256) void chirp_premul_kernel(const float2* __restrict__ X, float2* __restrict__ A, const float2* __restrict__ chirp,
                                                           int64_t N, int64_t L, int64_t M) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    float2 v = make_float2(0.f, 0.f);
    if (k < L) {
        const float2* xr = X + (int64_t)blockIdx.y * N;
        const bool shorter = L < N;
        const int64_t use = shorter ? L : N;
        const int64_t half = use / 2;
        const int64_t kk = k <= L / 2 ? k : L - k;
        if (kk <= half) {
            v = xr[kk];
            if (kk == half && (use % 2 == 0)) {
                const float f = shorter ? 2.f : 0.5f;
                v.x *= f;
                v.y *= f;
            }
            if (kk == 0 || (kk == L / 2 && L % 2 == 0)) v.y = 0.f;
            if (k > L / 2) v.y = -v.y;
        }
        const float2 c = chirp[k];
        v = make_float2(v.x * c.x - v.y * c.y, v.x * c.y + v.y * c.x);
    }
    A[(int64_t)blockIdx.y * M + k] = v;
}

__global__ __launch_bounds__(256) void pointwise_mul_kernel(float2* __restrict__ a, const float2* __restrict__ fb, int64_t M) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= M) return;
    float2* p = a + (int64_t)blockIdx.y * M + k;
    const float2 v = *p, w = fb[k];
    *p = make_float2(v.x * w.x - v.y * w.y, v.x * w.y + v.y * w.x);
}

__global__ __launch_bounds__(256) void chirp_trim_kernel(const float2* __restrict__ z, const float2* __restrict__ chirp, float* __restrict__ y,
                                                         int64_t ldy, int64_t M, int64_t rem0, int64_t final_len, float scale) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= final_len) return;
    const float2 v = z[(int64_t)blockIdx.y * M + rem0 + i], c = chirp[rem0 + i];
    y[(int64_t)blockIdx.y * ldy + i] = (v.x * c.x - v.y * c.y) * scale;
}

template <bool INV>
int run_fft(float2*& cur, float2*& other, int64_t N, int64_t rows, hipStream_t st) {
    int64_t Ns = 1;
    while (Ns < N) {
        const int64_t left = N / Ns;
        const int R = left >= 16 ? 16 : (int)left;
        const dim3 grid((unsigned)((N / R + 255) / 256), (unsigned)rows);
        if (R == 16) hipLaunchKernelGGL((stockham_kernel<16, INV>), grid, dim3(256), 0, st, cur, other, N, Ns);
        else if (R == 8) hipLaunchKernelGGL((stockham_kernel<8, INV>), grid, dim3(256), 0, st, cur, other, N, Ns);
        else if (R == 4) hipLaunchKernelGGL((stockham_kernel<4, INV>), grid, dim3(256), 0, st, cur, other, N, Ns);
        else hipLaunchKernelGGL((stockham_kernel<2, INV>), grid, dim3(256), 0, st, cur, other, N, Ns);
        int rc = lbm::check_launch("lbm_resample_fft_f32(fft pass)");
        if (rc != LBM_OK) return rc;
        float2* t = cur;
        cur = other;
        other = t;
        Ns *= R;
    }
    return LBM_OK;
}

}  // namespace

extern "C" int64_t lbm_resample_length(int64_t T, int32_t up, int32_t down) {
    const Plan p = make_plan(T, up, down);
    return p.ok ? p.final_len : 0;
}

extern "C" size_t lbm_resample_workspace_bytes(int64_t rows, int64_t T, int32_t up, int32_t down) {
    const Plan p = make_plan(T, up, down);
    if (!p.ok || rows < 1) return 0;
    if (p.M == 0) {
        const int64_t big = p.N > p.Nn ? p.N : p.Nn;
        return (size_t)rows * (size_t)big * sizeof(float2) * 2;
    }
    const int64_t big = p.N > p.M ? p.N : p.M;
    return ((size_t)rows * (size_t)big * 2 + (size_t)(2 * p.M + p.Nn)) * sizeof(float2);   // + kernel spectrum ping-pong + chirp
}

extern "C" int lbm_resample_fft_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t T, int32_t up,
                                    int32_t down, void* workspace, lbm_stream_t stream) {
    LBM_REQUIRE(rows >= 0 && T >= 0, LBM_EINVAL, "lbm_resample_fft_f32: negative size");
    if (rows == 0 || T == 0) return LBM_OK;
    LBM_REQUIRE(x && y && workspace, LBM_EINVAL, "lbm_resample_fft_f32: NULL pointer");
    LBM_REQUIRE(rows <= 65535, LBM_EUNSUPPORTED, "lbm_resample_fft_f32: at most 65535 rows per call");
    const Plan p = make_plan(T, up, down);
    LBM_REQUIRE(p.ok, LBM_EUNSUPPORTED, "lbm_resample_fft_f32: ratio %d/%d of a %lld-sample row is not supported (row too short or too long)",
                up, down, (long long)T);
    LBM_REQUIRE(ldx >= T && ldy >= p.final_len, LBM_EINVAL, "lbm_resample_fft_f32: row pitch shorter than the row");
    LBM_REQUIRE(((uintptr_t)workspace & 7) == 0, LBM_EINVAL, "lbm_resample_fft_f32: workspace must be 8-byte aligned");
    hipStream_t st = lbm::as_stream(stream);
    if (p.M != 0) {
        // resampled padded length is not a power of two: forward FFT as below, inverse by chirp-z convolution
        const int64_t big = p.N > p.M ? p.N : p.M;
        float2* a = static_cast<float2*>(workspace);
        float2* b = a + rows * big;
        float2* kb0 = b + rows * big;          // [M] convolution kernel / its spectrum (ping-pong with kb1)
        float2* kb1 = kb0 + p.M;
        float2* chirp = kb1 + p.M;             // [Nn]
        hipLaunchKernelGGL(chirp_kernel, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st, chirp, kb0, p.Nn, p.M);
        int rc = lbm::check_launch("lbm_resample_fft_f32(chirp)");
        if (rc != LBM_OK) return rc;
        float2 *kc = kb0, *ko = kb1;
        rc = run_fft<false>(kc, ko, p.M, 1, st);
        if (rc != LBM_OK) return rc;
        hipLaunchKernelGGL(resample_pad_kernel, dim3((unsigned)((p.N + 255) / 256), (unsigned)rows), dim3(256), 0, st, x, ldx, a, T, p.N, p.pad0);
        rc = lbm::check_launch("lbm_resample_fft_f32(pad)");
        if (rc != LBM_OK) return rc;
        float2 *cur = a, *other = b;
        rc = run_fft<false>(cur, other, p.N, rows, st);
        if (rc != LBM_OK) return rc;
        hipLaunchKernelGGL(chirp_premul_kernel, dim3((unsigned)((p.M + 255) / 256), (unsigned)rows), dim3(256), 0, st, cur, other, chirp, p.N,
                           p.Nn, p.M);
        rc = lbm::check_launch("lbm_resample_fft_f32(premul)");
        if (rc != LBM_OK) return rc;
        { float2* t = cur; cur = other; other = t; }
        rc = run_fft<false>(cur, other, p.M, rows, st);
        if (rc != LBM_OK) return rc;
        hipLaunchKernelGGL(pointwise_mul_kernel, dim3((unsigned)((p.M + 255) / 256), (unsigned)rows), dim3(256), 0, st, cur, kc, p.M);
        rc = lbm::check_launch("lbm_resample_fft_f32(mul)");
        if (rc != LBM_OK) return rc;
        rc = run_fft<true>(cur, other, p.M, rows, st);
        if (rc != LBM_OK) return rc;
        // 1/M of the convolution's inverse FFT, and irfft's 1/new_len times MNE's ratio new_len / orig_len = 1 / N
        hipLaunchKernelGGL(chirp_trim_kernel, dim3((unsigned)((p.final_len + 255) / 256), (unsigned)rows), dim3(256), 0, st, cur, chirp, y, ldy,
                           p.M, p.rem0, p.final_len, (float)(1.0 / ((double)p.N * (double)p.M)));
        return lbm::check_launch("lbm_resample_fft_f32(trim)");
    }
    const int64_t big = p.N > p.Nn ? p.N : p.Nn;
    float2* a = static_cast<float2*>(workspace);
    float2* b = a + rows * big;
    hipLaunchKernelGGL(resample_pad_kernel, dim3((unsigned)((p.N + 255) / 256), (unsigned)rows), dim3(256), 0, st, x, ldx, a, T, p.N, p.pad0);
    int rc = lbm::check_launch("lbm_resample_fft_f32(pad)");
    if (rc != LBM_OK) return rc;
    float2 *cur = a, *other = b;
    rc = run_fft<false>(cur, other, p.N, rows, st);
    if (rc != LBM_OK) return rc;
    hipLaunchKernelGGL(resample_spectrum_kernel, dim3((unsigned)((p.Nn + 255) / 256), (unsigned)rows), dim3(256), 0, st, cur, other, p.N, p.Nn);
    rc = lbm::check_launch("lbm_resample_fft_f32(spectrum)");
    if (rc != LBM_OK) return rc;
    {
        float2* t = cur;
        cur = other;
        other = t;
    }
    rc = run_fft<true>(cur, other, p.Nn, rows, st);
    if (rc != LBM_OK) return rc;
    // irfft's 1/new_len and MNE's ratio = new_len / orig_len combine to 1 / N
    hipLaunchKernelGGL(resample_trim_kernel, dim3((unsigned)((p.final_len + 255) / 256), (unsigned)rows), dim3(256), 0, st, cur, y, ldy, p.Nn,
                       p.rem0, p.final_len, (float)(1.0 / (double)p.N));
    return lbm::check_launch("lbm_resample_fft_f32(trim)");
}
