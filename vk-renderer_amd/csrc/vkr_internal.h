// Shared by the translation units that implement the C ABI: the context object, error reporting.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vkr_c.h"
#include "vkr_kernels.h"

struct vkr_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipDeviceProp_t props{};
  // Radius-cutoff resolve with light lists: coarse bins, per-tile lists and the lit-tile queue live in one
  // allocation that grows on demand and is freed with the context; the two alternating queue counters in another.
  void* tiled_ws = nullptr;
  size_t tiled_ws_bytes = 0;
  uint32_t* tile_counters = nullptr;
  uint32_t tile_epoch = 0;           // tiled launches made so far
  // Pinned host word into which every tiled call's strips launch stores the number of deep tiles of its frame (nobody
  // waits for it): by the next call it usually holds the previous frame's figure, which decides whether that call
  // launches the deep-tile kernel at all.  0xFFFFFFFF: nothing known yet.
  uint32_t* deep_hint = nullptr;
  uint32_t* deep_hint_device = nullptr;  // the same word as the device addresses it
  TiledLayout last_tiled{};          // layout, extent and light count of the most recent one (vkr_deferred_tile_stats)
  uint32_t last_width = 0, last_height = 0, last_lights = 0;
  // Single-launch cull: ticket counters + status words (vkr_cull_onepass_bytes), zeroed when (re)allocated and
  // when the sequence number wraps.
  void* cull_state = nullptr;
  size_t cull_state_bytes = 0;
  uint32_t cull_seq = 0;
  // Pinned host word that a deterministic cull sets when a workgroup abandoned its wait (kCullPatienceSeconds): the
  // results of that launch are wrong.  Allocated with the context; checked, reported and cleared by every library
  // call that waits for the stream.
  uint32_t* cull_fault = nullptr;
  uint32_t* cull_fault_device = nullptr;  // the same word as the device addresses it
};

// After a wait for the context's stream: VKR_ERR_DEVICE once if a deterministic cull since the last check gave up.
int32_t vkr_check_cull_fault(vkr_context* ctx);

// Records the message returned by vkr_last_error() for the calling thread and returns `code`.
int32_t vkr_fail(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define VKR_HIP(expr)                                                                                  \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return vkr_fail(e_ == hipErrorOutOfMemory ? VKR_ERR_OUT_OF_MEMORY : VKR_ERR_DEVICE, "%s: %s", #expr, \
                      hipGetErrorString(e_));                                                          \
  } while (0)
