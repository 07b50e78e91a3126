// Internal: the weights-stationary kernels of conv_wino_ws.hip, conv_k4s2_ws.hip and conv_k4s2_wino_ws.hip, reached through the entry points in conv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lbm {

bool wino3_ws_eligible(const float* x, int64_t ldx, const float* y, int64_t ldy, int Cin, int Cout);
int launch_wino3_ws(const float* x, const float* utab, const float* bias, float* y, int64_t B, int Lin, int64_t ldx, int Cout, int64_t ldy,
                    int relu, hipStream_t st);

// conv_k4s2_ws.hip: the encoder's 4-tap stride-2 layers (30 -> 64, 64 -> 128), reached through lbm_conv1d_pitched_f32
bool k4s2_ws_eligible(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, int Cin, int Cout, int Lin);
int launch_k4s2_ws(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout, int Lout,
                   int64_t ldy, int relu, hipStream_t st);

// conv_k4s2_wino_ws.hip: the same layers as two 2-tap phases in the minimal form F(3,2), reached through lbm_conv1d_k4s2_wino_f32
bool k4s2_wino_eligible(const float* x, int64_t ldx, const float* w, const float* y, int64_t ldy, int Cin, int Cout, int Lin);
int launch_k4s2_wino(const float* x, const float* w, const float* bias, float* y, int64_t B, int Cin, int Lin, int64_t ldx, int Cout, int Lout,
                     int64_t ldy, int relu, hipStream_t st);

}  // namespace lbm
