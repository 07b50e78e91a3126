// Internal: the weights-stationary Winograd kernel of conv_wino_ws.hip, reached through lbm_conv1d_wino3_f32 (conv.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace lbm {

bool wino3_ws_eligible(const float* x, int64_t ldx, const float* y, int64_t ldy, int Cin, int Cout);
int launch_wino3_ws(const float* x, const float* utab, const float* bias, float* y, int64_t B, int Lin, int64_t ldx, int Cout, int64_t ldy,
                    int relu, hipStream_t st);

}  // namespace lbm
