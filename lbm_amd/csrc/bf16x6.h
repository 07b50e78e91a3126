// Shared pieces of the bf16-piece ("bf16x6") matrix kernels: tile geometry, LDS image addressing, the fp32 -> three
// bf16 pieces split.  See vq_bf16.hip for the arithmetic and its accuracy.
#pragma once
#include "common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kRows = 128;        // input rows per workgroup
constexpr int kSup = 256;         // codes per super-tile
constexpr int kKC = 16;           // k per stage = one MFMA
constexpr int kCodePlane = kSup * 32, kXPlane = kRows * 32;            // bytes per piece plane
constexpr int kStageBytes = 3 * kCodePlane + 3 * kXPlane;              // 36864
constexpr int kLdsBytes = 2 * kStageBytes + 2 * kSup * 4 + 2 * kRows * 4 + 2 * kRows * 8;

__device__ __forceinline__ int swz(int row, int half) { return row * 32 + ((half ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ void split8(const float* v, bf16x8& p1, bf16x8& p2, bf16x8& p3) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h1 = (__bf16)v[e];
        const float r1 = v[e] - (float)h1;
        const __bf16 h2 = (__bf16)r1;
        const float r2 = r1 - (float)h2;
        p1[e] = h1;
        p2[e] = h2;
        p3[e] = (__bf16)r2;
    }
}

// Pre-split codebook table of the quantizer kernels:
//   [nst][3][Kp][16] bf16 pieces in k-stages of 16 (the six-product kernel's LDS image order),
//   float en[Kp]  = an upper bound of |e_k| (0 beyond K)                      -- the screening kernel's error bound
//   [Kp][2][256] bf16 = per code its first two pieces over all (zero padded) dims, 1 KiB per code   -- the screening kernel's tiles
struct VqTabPlan {
    int Kp, nst;
    size_t piece_bytes, rows_off, bytes;
};
inline VqTabPlan vq_tab_plan(int K, int D) {
    VqTabPlan p;
    p.Kp = (K + kSup - 1) / kSup * kSup;
    p.nst = (D + kKC - 1) / kKC;
    p.piece_bytes = (size_t)p.nst * 3 * p.Kp * 32;
    p.rows_off = p.piece_bytes + (size_t)p.Kp * sizeof(float);
    p.bytes = p.rows_off + (size_t)p.Kp * 1024;
    return p;
}

}  // namespace
