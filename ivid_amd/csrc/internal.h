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
