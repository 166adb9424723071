// Shared by the translation units that implement the C ABI: the context object, error reporting.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vkr_c.h"

struct vkr_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t props{};
  void* light_bins = nullptr;  // coarse light bins of the tiled resolve; grows on demand, freed with the context
  size_t light_bins_bytes = 0;
};

// Records the message returned by vkr_last_error() for the calling thread and returns `code`.
int32_t vkr_fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define VKR_HIP(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return vkr_fail(e_ == hipErrorOutOfMemory ? VKR_ERR_OUT_OF_MEMORY : VKR_ERR_DEVICE, "%s: %s", #expr, \
                      hipGetErrorString(e_));                                                          \
  } while (0)
