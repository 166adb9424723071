// Internal interface between the C-ABI layer (vkr_api.hip) and the gfx950 kernels.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

enum : uint32_t { kOutRgba8 = 0, kOutRgba16f = 1, kOutRgba32f = 2 };

// Kernel argument block of the deferred resolve (passed by value: lives in SGPRs / kernarg).
struct DeferredParams {
  const uint8_t* albedo;
  const uint8_t* material;
  const uint8_t* normal;
  const uint8_t* depth;
  uint8_t* out;
  const void* lights;  // 48-byte records: pos.xyz @0, color.xyz @16
  uint32_t pitch_albedo, pitch_material, pitch_normal, pitch_depth, pitch_out;
  uint32_t width, height;      // extent of the images handed to this launch
  uint32_t row_offset;         // first image row's index in the full frame
  uint32_t full_height;        // height of the full frame (UV denominator)
  uint32_t light_count;
  uint32_t out_mode;
  uint32_t flags;              // bit 0: radius cutoff
  float cam[3];
  float radius2;
  float inv_vp[16];            // column-major
  // Tile light lists, optional coarse level: for every kBinPixels x kBinPixels pixel bin the ascending
  // indices of the lights whose sphere of influence can reach the bin's view pyramid.  bin_list == null:
  // every tile tests the whole table.
  const uint32_t* bin_list;    // bins_x * bins_y * light_count entries, bin-major
  const uint32_t* bin_count;   // entries used per bin
  uint32_t bins_x;
};

constexpr uint32_t kBinPixels = 128;  // 4 x 4 tiles

// Entries the coarse light bins of a launch need (0 when the launch does not use them).
uint64_t vkr_deferred_bin_entries(const DeferredParams& p);
// bins: device memory for vkr_deferred_bin_entries(p) + bins_x * bins_y uint32 words, or null.
hipError_t vkr_launch_deferred(const DeferredParams& p, uint32_t* bins, hipStream_t stream);
hipError_t vkr_launch_exact_div_check(uint32_t n, uint32_t seed, uint32_t* mismatches, hipStream_t stream);

struct DrawCmd {
  uint32_t indexCount, instanceCount, firstIndex;
  int32_t vertexOffset;
  uint32_t firstInstance;
};
static_assert(sizeof(DrawCmd) == 20, "VkDrawIndexedIndirectCommand is 20 bytes");

struct CullParams {
  const float4* aabb_min;
  const float4* aabb_max;
  const uint32_t* mesh_base;
  const DrawCmd* cmd_template;
  DrawCmd* cmd_out;
  uint32_t* visible;
  uint32_t* error_flag;  // may be null
  void* workspace;       // deterministic mode
  uint32_t instance_count;
  uint32_t mesh_count;
  uint32_t first_mesh;
  uint32_t first_instance;
  float planes[24];
};

uint64_t vkr_cull_workspace_size(uint64_t instance_count, uint32_t mesh_count);
// The deterministic mode's workspace as a per-GPU array inside the code object, for the entry point that
// allocates nothing.  vkr_cull_static_init(device) resolves the array's address on `device`, which must be
// the current device (once per device, at context creation: the first use of a code object on a device
// may allocate); vkr_cull_static_workspace returns that address (null before init) and the array's size.
constexpr uint64_t kCullStaticMaxInstances = 1ull << 22;
constexpr uint32_t kCullStaticMaxMeshes = 1u << 16;
hipError_t vkr_cull_static_init(int device);
void* vkr_cull_static_workspace(int device, uint64_t* out_bytes);
hipError_t vkr_launch_cull_atomic(const CullParams& p, hipStream_t stream);
hipError_t vkr_launch_cull_deterministic(const CullParams& p, hipStream_t stream);
