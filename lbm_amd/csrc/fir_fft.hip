// Zero-phase FIR over [rows, T] (a1/a2) by fast convolution: the same y as fir.hip's direct form,
// computed with an LDS-staged in-register FFT (MNE itself applies the filter by FFT overlap-add).
//
// One row is cut into an even number of output blocks of Lb samples; two blocks a, b ride one complex
// transform z = a + i b of N = 256*R3 points (N >= Lb + ntaps - 1).  Because the taps are real,
// IFFT(FFT(z) * H) = (a * h) + i (b * h): no untangling pass.  The transform is 16 x 16 x R3:
//   stage 1  thread j holds z[j + Q n1] (Q = N/16), 16-point DFT over n1, twiddle w_N^{j k1}
//   stage 2  thread (k1, n3): 16-point DFT over n2, twiddle w_Q^{n3 k2}
//   stage 3  16/R3 R3-point DFTs per thread over n3  -> Z[k1 + 16 k2 + 256 k3]
// then the product with H (kept in registers, pre-permuted to the stage-3 register order and scaled by
// 1/N), and the three stages undone in reverse with conjugate twiddles -- the forward output order never
// has to be un-permuted.  Stages exchange data through one padded LDS buffer (conflict-free 8-byte
// accesses); stage-1 twiddles sit in LDS as [k1][j].  Workgroups are persistent: tables are loaded once,
// then (row, block pair) items are taken with a grid stride.  N/16 threads, 16 complex points per thread.
#include "common.h"
#include "fft_small.h"

namespace {

struct FftPlan {
    int R3;       // third radix: 4, 8 or 16  (N = 256 * R3)
    int N;
    int npairs;   // block pairs per row
    int Lb;       // outputs per block
};

// Cheapest N in {1024, 2048, 4096} by pairs * N * log2 N.
bool make_plan(int T, int ntaps, FftPlan* out) {
    double best = 0.0;
    bool found = false;
    for (int lg = 10; lg <= 12; ++lg) {
        const int N = 1 << lg;
        const int Lmax = N - (ntaps - 1);
        if (Lmax < 1) continue;
        const int nblk = (T + Lmax - 1) / Lmax;
        const int npairs = (nblk + 1) / 2;
        const double cost = (double)npairs * N * lg;
        if (!found || cost < best) {
            best = cost;
            found = true;
            out->R3 = N / 256;
            out->N = N;
            out->npairs = npairs;
            out->Lb = (T + 2 * npairs - 1) / (2 * npairs);
        }
    }
    return found;
}

// tables: float2 [N] Hq | [16][Q] stage-1 twiddles | [16][R3] stage-2 twiddles
template <int R3>
__global__ void fir_fft_tables_kernel(float2* __restrict__ tab, const float* __restrict__ taps, int ntaps) {
    constexpr int Q = 16 * R3, N = 16 * Q, NT = Q;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= 2 * N + Q) return;
    if (e < N) {
        const int i = e / NT, t = e % NT;
        const int g = i / R3, k3 = i % R3;
        const int u = g * NT + t;
        const int k = (u >> 4) + 16 * (u & 15) + 256 * k3;
        double sr = 0.0, si = 0.0;
        for (int n = 0; n < ntaps; ++n) {
            const int m = (int)(((long long)k * n) % N);
            double s, c;
            sincospi(2.0 * (double)m / (double)N, &s, &c);
            sr += (double)taps[n] * c;
            si -= (double)taps[n] * s;
        }
        tab[e] = make_float2((float)(sr / N), (float)(si / N));
    } else if (e < 2 * N) {
        const int k1 = (e - N) / Q, j = (e - N) % Q;
        double s, c;
        sincospi(2.0 * (double)((k1 * j) % N) / (double)N, &s, &c);
        tab[e] = make_float2((float)c, (float)-s);
    } else {
        const int k2 = (e - 2 * N) / R3, n3 = (e - 2 * N) % R3;
        double s, c;
        sincospi(2.0 * (double)((k2 * n3) % Q) / (double)Q, &s, &c);
        tab[e] = make_float2((float)c, (float)-s);
    }
}

// NSUB independent items per workgroup (NSUB * N/16 threads): the sub-groups share the stage-1 twiddle table
// and the barriers, which raises the waves per CU that fit next to the LDS (12 instead of 8 at N = 2048).
// NPF: 16-byte quads per thread that cover one window's real samples (min(T, Lb + N) / 4 / (N/16), <= 8).
template <int R3, int NPF, int NSUB, bool VEC>
__global__ __launch_bounds__(16 * R3 * NSUB, (R3 == 16 ? 2 : 3)) void fir_fft_kernel(const float* __restrict__ x, float* __restrict__ y, int T,
                                                          int ntaps, int pad_mode, int npairs, int Lb, int items,
                                                          const float2* __restrict__ tab) {
    constexpr int Q = 16 * R3, N = 16 * Q, NT = Q, G = 16 / R3;
    constexpr int A1S = Q + R3;          // stage-1 <-> stage-2 exchange: element (k1, j) at k1*A1S + j
    constexpr int A2S = 272 + 16 / R3;   // stage-2 <-> stage-3 exchange: element (k1, k2, n3) at n3*A2S + 17 k1 + k2
    extern __shared__ __align__(16) float2 fft_smem[];
    float2* s_t1 = fft_smem;             // [k1][j]  w_N^{j k1}
    float2* s_t2 = s_t1 + N;             // [k2][n3] w_Q^{n3 k2}
    constexpr int XCH = (16 * A1S > R3 * A2S) ? 16 * A1S : R3 * A2S;
    const int t = threadIdx.x % NT;
    const int sub = __builtin_amdgcn_readfirstlane(threadIdx.x / NT);   // NT is a multiple of 64: wave-uniform, so the
                                                                         // item arithmetic below stays on the scalar unit
    float2* s_x = s_t2 + Q + sub * XCH;  // [XCH] per sub-group

    for (int i = threadIdx.x; i < N; i += NT * NSUB) s_t1[i] = tab[N + i];
    for (int i = threadIdx.x; i < Q; i += NT * NSUB) s_t2[i] = tab[2 * N + i];
    float hre[16], him[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float2 h = tab[i * NT + t];
        hre[i] = h.x;
        him[i] = h.y;
    }
    __syncthreads();

    const int half = (ntaps - 1) >> 1;
    const bool reflect = pad_mode == LBM_PAD_REFLECT_LIMITED;
    const int seg = Lb + ntaps - 1;      // input samples a block needs (<= N)
    const int k1 = t / R3, n3 = t % R3;  // stage-2 role of this thread

    // Input staging.  The item's window of the padded row -- padded indices [a0, a0 + Lb + N), i.e. samples
    // ms = a0 - half onward -- is assembled in LDS (aliasing the exchange buffer): the real samples are copied
    // in with 16-byte loads issued one item ahead (registers `pre`), the reflected / zero padding is derived
    // from them in LDS, and then every thread picks its 2 x 16 transform inputs with plain ds_reads.  No
    // per-sample address arithmetic or selects in the transform path; inputs past Lb + ntaps - 1 only reach
    // discarded outputs, so they just have to be finite (they are written as 0).
    float* const P = reinterpret_cast<float*>(s_x);
    const int WL = Lb + N;               // LDS floats a transform pair reads
    float4 pre[NPF];
    auto window_start = [&](int item) { return 2 * (int)((unsigned)item % (unsigned)npairs) * Lb - half; };
    auto prefetch = [&](int item) {
        const float* xr = x + (long long)((unsigned)item / (unsigned)npairs) * T;
        const int ms = window_start(item);
        const int q0 = ms < 0 ? 0 : (ms & ~3);              // first sample of quad 0 (16-byte aligned when vec)
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            const int m = q0 + 4 * (t + NT * i);
            if constexpr (VEC) {
                pre[i] = *reinterpret_cast<const float4*>(xr + (m + 3 < T ? m : 0));
            } else {
                pre[i] = make_float4(xr[m < T ? m : 0], xr[m + 1 < T ? m + 1 : 0], xr[m + 2 < T ? m + 2 : 0],
                                     xr[m + 3 < T ? m + 3 : 0]);
            }
        }
    };
    // every sub-group runs every iteration (shared barriers); one past the end re-does the last item, stores off
    const int first = blockIdx.x * NSUB + sub;
    prefetch(first < items ? first : items - 1);

    for (int base = blockIdx.x * NSUB; base < items; base += gridDim.x * NSUB) {   // items + grid * NSUB < 2^31 (launcher)
        const bool active = base + sub < items;
        const int item = active ? base + sub : items - 1;
        const long long row = (unsigned)item / (unsigned)npairs;
        const int pair = (int)((unsigned)item % (unsigned)npairs);
        const int a0 = 2 * pair * Lb, b0 = a0 + Lb;
        const bool has_b = b0 < T;
        const int ms = a0 - half;
        const int q0 = ms < 0 ? 0 : (ms & ~3);

        // ---- real samples -> LDS (sample m sits at P[m - ms]); vec: T % 4 == 0, so a quad never straddles the row end
        __syncthreads();   // the previous item's last exchange reads are done
        const bool st4 = VEC && !(ms & 3);
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            float4 v = pre[i];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));   // pins the loads (see conv.hip)
            const int m = q0 + 4 * (t + NT * i);
            const int l = m - ms;
            if (st4 && l + 3 < WL) {
                if (m < T) *reinterpret_cast<float4*>(P + l) = v;
            } else {
                const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (m + c < T && l + c >= 0 && l + c < WL) P[l + c] = e[c];
            }
        }
        __syncthreads();
        // ---- padding, derived from the staged samples: left of sample 0, right of sample T-1
        {
            const int nl = ms < 0 ? (-ms < WL ? -ms : WL) : 0;
            const float x0 = nl ? P[-ms] : 0.f;
            for (int l = t; l < nl; l += NT) {
                const int mm = -(ms + l);                    // reflect about sample 0
                P[l] = (reflect && mm <= T - 1 && mm - ms < WL) ? 2.f * x0 - P[mm - ms] : 0.f;
            }
            const int r0 = T - ms;                           // first LDS slot right of the row (>= 1)
            const float xl = r0 < WL ? P[r0 - 1] : 0.f;
            for (int l = r0 + t; l < WL; l += NT) {
                const int mm = 2 * (T - 1) - (ms + l);       // reflect about sample T-1
                const bool live = l < Lb + seg;
                P[l] = (reflect && live && mm >= 0 && mm >= ms) ? 2.f * xl - P[mm - ms] : 0.f;
            }
        }
        __syncthreads();
        float re[16], im[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int p = t + Q * n1;
            re[n1] = P[p];
            im[n1] = has_b ? P[Lb + p] : 0.f;
        }
        __syncthreads();   // the window is consumed; s_x becomes the exchange buffer
        {   // next item's loads fly under this item's transforms (the last item re-reads itself, unused)
            const int nxt = base + gridDim.x * NSUB + sub;
            prefetch(nxt < items ? nxt : items - 1);
        }

        // ---- forward stage 1
        lbm::dft16(re, im);
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float2 w = s_t1[k * Q + t];
            const float a = re[k], b = im[k];
            re[k] = a * w.x - b * w.y;
            im[k] = a * w.y + b * w.x;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s_x[k * A1S + t] = make_float2(re[k], im[k]);
        __syncthreads();
        // ---- forward stage 2
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) {
            const float2 v = s_x[k1 * A1S + n2 * R3 + n3];
            re[n2] = v.x;
            im[n2] = v.y;
        }
        __syncthreads();   // s_x is reused with the stage-3 layout
        lbm::dft16(re, im);
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float2 w = s_t2[k * R3 + n3];
            const float a = re[k], b = im[k];
            re[k] = a * w.x - b * w.y;
            im[k] = a * w.y + b * w.x;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s_x[n3 * A2S + 17 * k1 + k] = make_float2(re[k], im[k]);
        __syncthreads();
        // ---- forward stage 3, product with H, inverse stage 3 (this thread re-writes exactly what it read)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int u = g * NT + t;
            const int base = u + (u >> 4);
#pragma unroll
            for (int n = 0; n < R3; ++n) {
                const float2 v = s_x[n * A2S + base];
                re[g * R3 + n] = v.x;
                im[g * R3 + n] = v.y;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) lbm::dft<R3>(re + g * R3, im + g * R3);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = re[i], b = im[i];
            re[i] = a * hre[i] - b * him[i];
            im[i] = a * him[i] + b * hre[i];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) lbm::dft<R3>(im + g * R3, re + g * R3);
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int u = g * NT + t;
            const int base = u + (u >> 4);
#pragma unroll
            for (int n = 0; n < R3; ++n) s_x[n * A2S + base] = make_float2(re[g * R3 + n], im[g * R3 + n]);
        }
        __syncthreads();
        // ---- inverse stage 2
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float2 v = s_x[n3 * A2S + 17 * k1 + k];
            re[k] = v.x;
            im[k] = v.y;
        }
        __syncthreads();
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float2 w = s_t2[k * R3 + n3];
            const float a = re[k], b = im[k];
            re[k] = a * w.x + b * w.y;
            im[k] = b * w.x - a * w.y;
        }
        lbm::dft16(im, re);
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) s_x[k1 * A1S + n2 * R3 + n3] = make_float2(re[n2], im[n2]);
        __syncthreads();
        // ---- inverse stage 1 (reads the slots this thread writes first in the next item: no barrier needed)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float2 v = s_x[k * A1S + t];
            re[k] = v.x;
            im[k] = v.y;
        }
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float2 w = s_t1[k * Q + t];
            const float a = re[k], b = im[k];
            re[k] = a * w.x + b * w.y;
            im[k] = b * w.x - a * w.y;
        }
        lbm::dft16(im, re);

        float* yr = y + row * (long long)T;
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
            const int o = t + Q * n1 - (ntaps - 1);
            if (active && o >= 0 && o < Lb) {
                if (a0 + o < T) yr[a0 + o] = re[n1];
                if (b0 + o < T) yr[b0 + o] = im[n1];
            }
        }
    }
}

template <int R3>
int launch_tables(float2* tab, const float* taps, int ntaps, hipStream_t s) {
    constexpr int total = 2 * (256 * R3) + 16 * R3;
    hipLaunchKernelGGL(fir_fft_tables_kernel<R3>, dim3((total + 255) / 256), dim3(256), 0, s, tab, taps, ntaps);
    return lbm::check_launch("lbm_fir_fft_tables_init");
}

template <int R3, int NPF, int NSUB>
int launch_fir_k(const float* x, float* y, int64_t rows, int T, int ntaps, int pad_mode, const FftPlan& p, const float2* tab,
                 hipStream_t s) {
    const long long items = (long long)rows * p.npairs;
    constexpr int Q = 16 * R3, N = 16 * Q;
    constexpr int xch = (16 * (Q + R3) > R3 * (272 + 16 / R3)) ? 16 * (Q + R3) : R3 * (272 + 16 / R3);
    constexpr size_t lds = (size_t)(N + Q + NSUB * xch) * sizeof(float2);
    const int per_cu = (int)((160 * 1024) / lds) < 8 ? (int)((160 * 1024) / lds) : 8;
    long long grid = 256LL * per_cu;
    const long long need = (items + NSUB - 1) / NSUB;
    if (grid > need) grid = need;
    const bool vec = (T % 4 == 0) && (((uintptr_t)x & 15) == 0);   // every row starts 16-byte aligned
    auto k = vec ? fir_fft_kernel<R3, NPF, NSUB, true> : fir_fft_kernel<R3, NPF, NSUB, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return lbm::fail(LBM_ELAUNCH, "lbm_fir_fft_f32: hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(16 * R3 * NSUB), lds, s, x, y, T, ntaps, pad_mode, p.npairs, p.Lb, (int)items, tab);
    return lbm::check_launch("lbm_fir_fft_f32");
}

template <int R3, int NSUB>
int launch_fir(const float* x, float* y, int64_t rows, int T, int ntaps, int pad_mode, const FftPlan& p, const float2* tab,
               hipStream_t s) {
    // quads per thread that cover the real samples of one window (+1 quad of alignment slack)
    const int real = T < p.Lb + p.N ? T : p.Lb + p.N;
    const int quads = (real + 3) / 4 + 1;
    if (quads <= 4 * 16 * R3) return launch_fir_k<R3, 4, NSUB>(x, y, rows, T, ntaps, pad_mode, p, tab, s);
    return launch_fir_k<R3, 8, NSUB>(x, y, rows, T, ntaps, pad_mode, p, tab, s);
}

}  // namespace

extern "C" size_t lbm_fir_fft_tables_bytes(int32_t T, int32_t ntaps) {
    FftPlan p;
    if (T < 1 || ntaps < 1 || !(ntaps & 1) || !make_plan(T, ntaps, &p)) return 0;
    return (size_t)(2 * p.N + p.N / 16) * sizeof(float2);
}

extern "C" int lbm_fir_fft_tables_init(void* tables, const float* taps, int32_t T, int32_t ntaps, lbm_stream_t stream) {
    LBM_REQUIRE(tables && taps, LBM_EINVAL, "lbm_fir_fft_tables_init: NULL pointer");
    LBM_REQUIRE(T >= 1, LBM_EINVAL, "lbm_fir_fft_tables_init: T must be >= 1 (got %d)", T);
    LBM_REQUIRE(ntaps >= 1 && (ntaps & 1), LBM_EINVAL, "lbm_fir_fft_tables_init: ntaps must be odd and >= 1 (got %d)", ntaps);
    LBM_REQUIRE(((uintptr_t)tables & 15) == 0, LBM_EINVAL, "lbm_fir_fft_tables_init: tables must be 16-byte aligned");
    FftPlan p;
    LBM_REQUIRE(make_plan(T, ntaps, &p), LBM_EUNSUPPORTED,
                "lbm_fir_fft_tables_init: ntaps=%d does not fit a 4096-point transform (use lbm_fir_f32)", ntaps);
    float2* tab = static_cast<float2*>(tables);
    hipStream_t s = lbm::as_stream(stream);
    if (p.R3 == 4) return launch_tables<4>(tab, taps, ntaps, s);
    if (p.R3 == 8) return launch_tables<8>(tab, taps, ntaps, s);
    return launch_tables<16>(tab, taps, ntaps, s);
}

extern "C" int lbm_fir_fft_f32(const float* x, float* y, int64_t rows, int32_t T, const void* tables, int32_t ntaps,
                               int32_t pad_mode, lbm_stream_t stream) {
    LBM_REQUIRE(rows >= 0 && T >= 0, LBM_EINVAL, "lbm_fir_fft_f32: negative size (rows=%lld, T=%d)", (long long)rows, T);
    if (rows == 0 || T == 0) return LBM_OK;
    LBM_REQUIRE(x && y && tables, LBM_EINVAL, "lbm_fir_fft_f32: NULL pointer");
    LBM_REQUIRE(ntaps >= 1 && (ntaps & 1), LBM_EINVAL, "lbm_fir_fft_f32: ntaps must be odd and >= 1 (got %d)", ntaps);
    LBM_REQUIRE(pad_mode == LBM_PAD_REFLECT_LIMITED || pad_mode == LBM_PAD_ZERO, LBM_EINVAL,
                "lbm_fir_fft_f32: unknown pad_mode %d", pad_mode);
    LBM_REQUIRE(((uintptr_t)x & 3) == 0 && ((uintptr_t)y & 3) == 0 && ((uintptr_t)tables & 15) == 0, LBM_EINVAL,
                "lbm_fir_fft_f32: misaligned pointer");
    FftPlan p;
    LBM_REQUIRE(make_plan(T, ntaps, &p), LBM_EUNSUPPORTED,
                "lbm_fir_fft_f32: ntaps=%d does not fit a 4096-point transform (use lbm_fir_f32)", ntaps);
    LBM_REQUIRE((long long)rows * p.npairs <= 0x7ff00000LL, LBM_EUNSUPPORTED, "lbm_fir_fft_f32: too many rows (rows x block pairs must stay below 2^31)");
    const float2* tab = static_cast<const float2*>(tables);
    hipStream_t s = lbm::as_stream(stream);
    if (p.R3 == 4) return launch_fir<4, 2>(x, y, rows, T, ntaps, pad_mode, p, tab, s);
    if (p.R3 == 8) return launch_fir<8, 2>(x, y, rows, T, ntaps, pad_mode, p, tab, s);
    return launch_fir<16, 1>(x, y, rows, T, ntaps, pad_mode, p, tab, s);
}
