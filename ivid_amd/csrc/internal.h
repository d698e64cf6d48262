// Host-side helpers shared by the translation units of libivid_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ivid_hip.h"

int ivid_set_error(const char* what, hipError_t e);  // records message, returns nonzero
int ivid_check_launch(const char* what);             // hipGetLastError() after a launch
const void* ivid_zero_page();                        // 256 zero bytes in device memory (per device)

// bytes per stored activation element of a dtype code (0 = invalid)
static inline int ivid_esz(int dtype) {
  return (dtype == IVID_F32 || dtype == IVID_BF16X3) ? 4 : ((dtype == IVID_BF16 || dtype == IVID_F16) ? 2 : 0);
}

// csrc/conv3x3_fused128.hip: the Cout <= 128 shape of the fused GroupNorm-apply + SiLU + conv3x3 kernel (conv3x3_fused_body.h)
bool ivid_fused128_supports(int dtype, int C0, int C1, int H, int W, int Cout, int skipC0, int skipC1);
int ivid_fused128_launch(int dtype, const void* src0, int C0, const void* src1, int C1, const float* ab, int up,
                         const void* weight, const float* bias, void* out, const void* res, int res_mode, int N, int H, int W,
                         int Cout, float* stats, const void* skip0, int skipC0, const void* skip1, int skipC1,
                         const void* skip_weight, void* stream, void* out_lo = nullptr, const void* res_lo = nullptr,
                         const void* src0_lo = nullptr, const void* src1_lo = nullptr, const void* skip0_lo = nullptr,
                         const void* skip1_lo = nullptr, const void* skip_weight_lo = nullptr, void* out16_hi = nullptr,
                         void* out16_lo = nullptr);
