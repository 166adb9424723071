// Internal interface between the C-ABI layer (vkr_api.hip) and the gfx950 kernels.
#pragma once

#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

enum : uint32_t { kOutRgba8 = 0, kOutRgba16f = 1, kOutRgba32f = 2, kOutBgra8 = 3 };

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
  uint32_t flags;              // bit 0: radius cutoff; bit 1: no light lists; bit 2: always four-wave workgroups;
                               // bit 3 (set by the launcher): no deep tiles this launch - count them, shade them as strips
  float cam[3];
  float radius2;
  float inv_vp[16];            // column-major
  // Radius cutoff with light lists.  Coarse level, optional: for every kBinPixels x kBinPixels pixel bin the
  // ascending indices of the lights whose sphere of influence can reach the bin's view pyramid
  // (bin_list == null: every tile tests the whole table).  Fine level: per 32 x 32 pixel tile the ascending
  // indices of the lights that can reach the tile, and the queue of tiles that have any.
  const uint32_t* bin_list;    // bins_x * bins_y * light_count entries, bin-major
  const uint32_t* bin_count;   // entries used per bin
  uint32_t bins_x;
  uint32_t tiles_x, tiles_y;   // ceil(width / 32), ceil(height / 32)
  uint32_t list_cap;           // entries per tile list
  const uint16_t* tile_list;   // tiles_x * tiles_y * list_cap
  const uint32_t* tile_count;  // per tile: list length; kTileOverflow set: the list holds only its first list_cap entries
  const uint32_t* lit_queue;   // kListClasses + 1 queues of tiles_x * tiles_y entries: the tiles with a non-empty list, by list class; the last one: the deep tiles
  const uint32_t* lit_counters; // [c] = length of queue c
  const uint16_t* tile_order;  // tiles_x * tiles_y * 1024: for tiles marked kTileDeep, their pixels (y << 5 | x within the tile) grouped by depth
  uint32_t* deep_hint;         // host memory (or null): the strips launch leaves the number of tiles that qualify as deep there
  const uint16_t* group_list;  // tiles_x * tiles_y * 8 * kGroupListCap: per group of 128 pixels of a deep tile, the entries of the tile's list that can reach it
  const uint16_t* group_count; // tiles_x * tiles_y * 8: their number; kGroupOverflow: more than a group list holds, the group walks the tile's list
};

constexpr uint32_t kBinPixels = 128;  // 4 x 4 tiles
constexpr uint32_t kTileOverflow = 0x80000000u;
constexpr uint32_t kTileDeep = 0x40000000u;     // the tile spans more depth than a light reaches: it is shaded in groups of pixels of similar depth
constexpr uint32_t kTileCountMask = 0x3FFFFFFFu;
constexpr uint32_t kGroupListCap = 128;         // entries per group list
constexpr uint32_t kGroupOverflow = 0xFFFFu;
constexpr uint32_t kListClasses = 4;         // queues of lit tiles, by list length
constexpr uint32_t kListCounterStride = 8;   // words per counter set: kListClasses queue lengths, the deep tiles' queue length, the number of tiles that qualify as deep, padding

// What tile_lists_kernel writes.  Queue lengths and ticket live in one of two counter sets that alternate from one
// tiled launch of a context to the next: a launch works on counters_now (zero on entry) and zeroes
// counters_next, which nothing reads until the following launch makes it its counters_now.
struct TileListOut {
  uint16_t* tile_list;
  uint32_t* tile_count;
  uint32_t* lit_queue;
  uint32_t* counters_now;
  uint32_t* counters_next;
  uint16_t* tile_order;
  uint16_t* group_list;
  uint16_t* group_count;
};

// Device memory a launch with the radius cutoff and light lists needs (tiled == false: none).
struct TiledLayout {
  bool tiled;
  uint32_t tiles_x, tiles_y, bins_x, list_cap;
  uint64_t bins, bin_entries;
  uint64_t off_bin_list, off_bin_count, off_tile_count, off_lit_queue, off_tile_list, off_tile_order, off_group_list, off_group_count, total_bytes;
};
TiledLayout vkr_deferred_tiled_layout(const DeferredParams& p);
// tiled_ws: total_bytes of 256-byte-aligned device memory when the layout says tiled, else ignored.
// counters: 2 x kListCounterStride device words, zero before the context's first tiled launch; epoch: number of tiled launches the
// context has made so far.  cu_count sizes the grid of the list-shading launch (workgroups that stay resident).
// deep_tiles_expected: whether to launch the kernel that shades deep tiles (see kTileDeep); false makes the listing
// launch treat every tile as an ordinary one.  Either way counters[set of this launch][kDeepCounter] ends up as the number of
// tiles that qualify.
constexpr uint32_t kDeepCounter = 5;
constexpr uint32_t kFlagNoDeepTiles = 8u;
hipError_t vkr_launch_deferred(const DeferredParams& p, void* tiled_ws, uint32_t* counters, uint32_t epoch, uint32_t cu_count,
                               bool deep_tiles_expected, uint32_t* deep_hint, hipStream_t stream);
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
  uint32_t instance_count;
  uint32_t mesh_count;
  uint32_t first_mesh;
  uint32_t first_instance;
  float planes[24];
};

// Deterministic mode: its state is one 64-bit status word per 1024 instances, all zero before the first call.
// seq: 1 .. kCullSeqEnd - 1, different from the seq of every earlier call on the same state since it was last
// zeroed.  resident_blocks: how many workgroups of the kernel the device holds at once
// (vkr_cull_blocks_per_cu() x CUs, or fewer); the grid never exceeds it.
// fault: a word in host memory (device address, may be null) that a launch sets to its seq when a workgroup
// abandons the wait for another one (after kCullPatienceSeconds: the results of that launch are wrong).
struct OnePassState {
  uint64_t* status;
  uint32_t* fault;
  uint64_t patience_ticks;  // of the device's wall clock
  uint32_t seq;
};
constexpr uint32_t kCullPatienceSeconds = 10;
constexpr uint32_t kCullSeqEnd = 1u << 30;
// A launch recorded into a HIP graph runs again with the arguments it was recorded with, sequence number
// included, so it cannot rely on "different from every earlier call": recorded launches all use this number,
// which the counters never hand out, and are preceded by a recorded clear of the state.
constexpr uint32_t kCullSeqRecorded = kCullSeqEnd - 1u;
uint64_t vkr_cull_state_bytes(uint64_t instance_count);
uint32_t vkr_cull_blocks_per_cu();
// commands_ready: cmd_out already holds the templates with zero counts (vkr_launch_cull_reset), skip that launch
hipError_t vkr_launch_cull_atomic(const CullParams& p, bool commands_ready, hipStream_t stream);
hipError_t vkr_launch_cull_reset(const CullParams& p, hipStream_t stream);
hipError_t vkr_launch_cull_deterministic(const CullParams& p, void* state_mem, uint32_t seq, uint32_t resident_blocks, uint32_t* fault, hipStream_t stream);

// The same state as a per-GPU array inside the code object, for the entry point that allocates nothing.
// vkr_cull_static_init(device) resolves the array's address on `device`, which must be the current device, and
// creates the event that orders the array's users (once per device, at context creation: the first use of a code
// object on a device may allocate).
// vkr_launch_cull_static runs the deterministic kernel in that array (hipErrorNotInitialized before init,
// hipErrorInvalidValue for more than kCullStaticMaxInstances).  All launches on a device share the array, whatever
// their context: each one is made to wait for the one before (an event recorded behind every launch, waited for
// in front of the next), so launches from different streams or host threads do not overlap.  On a stream that is
// being captured into a HIP graph the launch is recorded with the sequence number reserved for recorded launches
// and a clear of the array in front of it, and takes no part in that ordering: replays that may overlap other
// users of the array on the same device are the caller's to order.
constexpr uint64_t kCullStaticMaxInstances = 1ull << 22;
hipError_t vkr_cull_static_init(int device);
hipError_t vkr_launch_cull_static(const CullParams& p, int device, uint32_t resident_blocks, uint32_t* fault, hipStream_t stream);
