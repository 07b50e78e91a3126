// In-register complex DFTs of 4, 8 and 16 points (split re/im arrays, natural order in and out).
// Forward sign e^{-2 pi i n k / R}.  The inverse (unnormalised) is the same routine with the re and im
// arrays exchanged:  idft(re, im) == dft(im, re).
// Every index is a compile-time constant after unrolling, so the arrays live in VGPRs.
#pragma once

#if defined(__HIPCC__)
#define LBM_HD __host__ __device__ __forceinline__
#else
#define LBM_HD inline
#endif

namespace lbm {

// 4-point butterfly on elements 0, S, 2S, 3S; element S*c receives X_c.
template <int S>
LBM_HD void bfly4(float* re, float* im) {
    const float ar = re[0] + re[2 * S], ai = im[0] + im[2 * S];
    const float br = re[0] - re[2 * S], bi = im[0] - im[2 * S];
    const float cr = re[S] + re[3 * S], ci = im[S] + im[3 * S];
    const float dr = re[S] - re[3 * S], di = im[S] - im[3 * S];
    re[0] = ar + cr;      im[0] = ai + ci;
    re[2 * S] = ar - cr;  im[2 * S] = ai - ci;
    re[S] = br + di;      im[S] = bi - dr;       // b - i d
    re[3 * S] = br - di;  im[3 * S] = bi + dr;   // b + i d
}

// (re, im) *= w16^m = exp(-2 pi i m / 16), m a compile-time constant after unrolling
LBM_HD void mul_w16(float& re, float& im, int m) {
    constexpr float kC1 = 0.92387953251128675613f;   // cos(pi/8)
    constexpr float kS1 = 0.38268343236508977173f;   // sin(pi/8)
    constexpr float kH = 0.70710678118654752440f;    // sqrt(1/2)
    const float a = re, b = im;
    switch (m & 15) {
        case 0: break;
        case 1: re = a * kC1 + b * kS1; im = b * kC1 - a * kS1; break;
        case 2: re = (a + b) * kH;      im = (b - a) * kH;      break;
        case 3: re = a * kS1 + b * kC1; im = b * kS1 - a * kC1; break;
        case 4: re = b;                 im = -a;                break;
        case 5: re = b * kC1 - a * kS1; im = -(a * kC1 + b * kS1); break;
        case 6: re = (b - a) * kH;      im = -(a + b) * kH;     break;
        case 7: re = b * kS1 - a * kC1; im = -(a * kS1 + b * kC1); break;
        case 8: re = -a;                im = -b;                break;
        case 9: re = -(a * kC1 + b * kS1); im = a * kS1 - b * kC1; break;
        default: {   // not reached by the 4x4 / 4x2 decompositions below (m <= 9)
            const float ang = -0.39269908169872415481f * (float)(m & 15);
            const float c = __builtin_cosf(ang), s = __builtin_sinf(ang);
            re = a * c - b * s; im = a * s + b * c;
        }
    }
}

LBM_HD void dft4(float* re, float* im) { bfly4<1>(re, im); }

// n = 2a + b, k = c + 4d
LBM_HD void dft8(float* re, float* im) {
    bfly4<2>(re, im);
    bfly4<2>(re + 1, im + 1);
#pragma unroll
    for (int c = 1; c < 4; ++c) mul_w16(re[2 * c + 1], im[2 * c + 1], 2 * c);   // w8^c
    float tr[8], ti[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        tr[c] = re[2 * c] + re[2 * c + 1];      ti[c] = im[2 * c] + im[2 * c + 1];
        tr[c + 4] = re[2 * c] - re[2 * c + 1];  ti[c + 4] = im[2 * c] - im[2 * c + 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { re[k] = tr[k]; im[k] = ti[k]; }
}

// n = 4a + b, k = c + 4d
LBM_HD void dft16(float* re, float* im) {
#pragma unroll
    for (int b = 0; b < 4; ++b) bfly4<4>(re + b, im + b);            // element 4c+b = Y[b][c]
#pragma unroll
    for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int b = 1; b < 4; ++b) mul_w16(re[4 * c + b], im[4 * c + b], b * c);
#pragma unroll
    for (int c = 0; c < 4; ++c) bfly4<1>(re + 4 * c, im + 4 * c);    // element 4c+d = X[c+4d]
    float tr[16], ti[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { tr[(q >> 2) + 4 * (q & 3)] = re[q]; ti[(q >> 2) + 4 * (q & 3)] = im[q]; }
#pragma unroll
    for (int k = 0; k < 16; ++k) { re[k] = tr[k]; im[k] = ti[k]; }
}

template <int R>
LBM_HD void dft(float* re, float* im) {
    if (R == 4) dft4(re, im);
    else if (R == 8) dft8(re, im);
    else dft16(re, im);
}

}  // namespace lbm
