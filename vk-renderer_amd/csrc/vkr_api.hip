// C ABI over HIP: contexts, buffers, the typed entry points and the command-list dispatch surface
// declared in include/vkr_c.h.  No exception leaves this file (the two places that can run out of host memory,
// the dispatch snapshot list and the statistics download, catch it); every failure sets the calling thread's
// error string, which lives in fixed storage, and returns a vkr_status.

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "vkr_c.h"
#include "vkr_internal.h"
#include "vkr_kernels.h"
#include "vkr_math.h"

namespace {

thread_local char g_last_error[512] = "";  // fixed storage: reporting a failure never allocates

#define fail vkr_fail

uint32_t format_texel_bytes(uint32_t f) {
  switch (f) {
    case VKR_FORMAT_R8G8B8A8_UNORM:
    case VKR_FORMAT_B8G8R8A8_UNORM:
    case VKR_FORMAT_R32_SFLOAT:
    case VKR_FORMAT_D32_SFLOAT:
      return 4;
    case VKR_FORMAT_R16G16B16A16_SFLOAT:
      return 8;
    case VKR_FORMAT_R32G32B32A32_SFLOAT:
      return 16;
    default:
      return 0;
  }
}

}  // namespace

struct vkr_buffer {
  vkr_context* ctx = nullptr;
  void* device = nullptr;
  void* host = nullptr;  // HOST_VISIBLE only
  uint64_t bytes = 0;
  bool owned = true;
};

enum KernelId : uint32_t { kKernelDeferred = 0, kKernelPrepIndirect = 1 };

struct vkr_kernel {
  KernelId id;
  const char* name;
  uint32_t push_min, push_max;
};

namespace {

vkr_kernel g_kernels[] = {
    {kKernelDeferred, "deferred", 80, 96},
    {kKernelPrepIndirect, "prep_indirect", 112, 112},
};

struct BoundBuffer {
  vkr_buffer* buf = nullptr;
  uint64_t offset = 0, range = 0;
};
struct BoundImage {
  vkr_buffer* buf = nullptr;
  uint64_t offset = 0;
  uint32_t width = 0, height = 0, pitch = 0, format = 0;
};

constexpr uint32_t kMaxSets = 2, kMaxBindings = 8, kMaxArray = 4;

struct BindState {
  vkr_kernel* kernel = nullptr;
  uint8_t push[128] = {};
  uint32_t push_bytes = 0;
  BoundBuffer buffers[kMaxSets][kMaxBindings];
  BoundImage images[kMaxSets][kMaxBindings][kMaxArray];
};

struct Dispatch {
  BindState state;  // snapshot at record time (Vulkan semantics: state is captured per dispatch)
  uint32_t x, y, z;
};

}  // namespace

struct vkr_cmd {
  vkr_context* ctx = nullptr;
  BindState cur;
  std::vector<Dispatch> dispatches;
  bool ended = false;
};

struct vkr_fence {
  vkr_context* ctx = nullptr;
  hipEvent_t event = nullptr;
};

namespace {

int32_t check_image(const vkr_image2d& im, const char* what, uint32_t want_a, uint32_t want_b, uint32_t want_c,
                    uint32_t w, uint32_t h) {
  if (!im.device) return fail(VKR_ERR_INVALID_ARGUMENT, "%s: null device pointer", what);
  if (im.format != want_a && im.format != want_b && im.format != want_c)
    return fail(VKR_ERR_INVALID_ARGUMENT, "%s: unsupported format %u", what, im.format);
  if (im.width != w || im.height != h)
    return fail(VKR_ERR_INVALID_ARGUMENT, "%s: extent %ux%u does not match the output extent %ux%u", what, im.width,
                im.height, w, h);
  const uint32_t texel = format_texel_bytes(im.format);
  // every lane reads/writes whole 4-texel groups, so a row must hold ceil(w/4)*4 texels
  const uint64_t need = (uint64_t)((w + 3u) / 4u) * 4u * texel;
  if (im.pitch_bytes < need)
    return fail(VKR_ERR_INVALID_ARGUMENT, "%s: pitch %u < %llu bytes (rows are accessed in 4-texel groups)", what,
                im.pitch_bytes, (unsigned long long)need);
  if ((im.pitch_bytes & 15u) != 0u || (reinterpret_cast<uintptr_t>(im.device) & 15u) != 0u)
    return fail(VKR_ERR_INVALID_ARGUMENT, "%s: rows must be 16-byte aligned (ptr %p, pitch %u)", what, im.device,
                im.pitch_bytes);
  return VKR_OK;
}

// A launch recorded into a HIP graph runs again with its recorded arguments; calls whose kernels take a value
// that the host advances per call (queue parity, sequence number) refuse to be recorded.
bool stream_is_recording(vkr_context* ctx) {
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(ctx->stream, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone;
}

int32_t run_deferred(vkr_context* ctx, const vkr_deferred_args* a) {
  if (!ctx || !a) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_deferred_resolve: null argument");
  const uint32_t w = a->out.width, h = a->out.height;
  if (w == 0u || h == 0u) return VKR_OK;  // empty dispatch
  int32_t rc;
  if ((rc = check_image(a->out, "out", VKR_FORMAT_R8G8B8A8_UNORM, VKR_FORMAT_R16G16B16A16_SFLOAT,
                        VKR_FORMAT_R32G32B32A32_SFLOAT, w, h)) != VKR_OK) {
    if (a->out.format != VKR_FORMAT_B8G8R8A8_UNORM) return rc;
    vkr_image2d o = a->out;
    o.format = VKR_FORMAT_R8G8B8A8_UNORM;
    if ((rc = check_image(o, "out", VKR_FORMAT_R8G8B8A8_UNORM, 0, 0, w, h)) != VKR_OK) return rc;
  }
  if ((rc = check_image(a->albedo, "albedo", VKR_FORMAT_R8G8B8A8_UNORM, 0, 0, w, h)) != VKR_OK) return rc;
  if ((rc = check_image(a->material, "material", VKR_FORMAT_R8G8B8A8_UNORM, 0, 0, w, h)) != VKR_OK) return rc;
  if ((rc = check_image(a->normal, "normal", VKR_FORMAT_R16G16B16A16_SFLOAT, 0, 0, w, h)) != VKR_OK) return rc;
  if ((rc = check_image(a->depth, "depth", VKR_FORMAT_D32_SFLOAT, VKR_FORMAT_R32_SFLOAT, 0, w, h)) != VKR_OK) return rc;
  if (a->light_count != 0u && !a->lights) return fail(VKR_ERR_INVALID_ARGUMENT, "lights: null with light_count %u", a->light_count);
  if (a->lights && (reinterpret_cast<uintptr_t>(a->lights) & 15u) != 0u)
    return fail(VKR_ERR_INVALID_ARGUMENT, "lights: must be 16-byte aligned");
  const uint32_t full_h = a->full_height ? a->full_height : h;
  if ((uint64_t)a->row_offset + h > full_h)
    return fail(VKR_ERR_INVALID_ARGUMENT, "row_offset %u + height %u exceeds full_height %u", a->row_offset, h, full_h);
  if (a->flags & ~(VKR_DEFERRED_RADIUS_CUTOFF | VKR_DEFERRED_NO_TILE_LISTS | VKR_DEFERRED_WHOLE_TILE_WORKGROUPS)) return fail(VKR_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", a->flags);
  if (a->flags & VKR_DEFERRED_RADIUS_CUTOFF) {
    const float r = a->push[3];
    // the in-kernel cutoff is the exact step clamp((r^2 - d^2) * 2^80): r^2 must stay within its exact range
    if (!(r >= 1e-8f && (r <= 1e7f || std::isinf(r)))) return fail(VKR_ERR_INVALID_ARGUMENT, "pointLightRadius %g outside [1e-8, 1e7] (or +inf)", (double)r);
  }

  DeferredParams p{};
  p.albedo = static_cast<const uint8_t*>(a->albedo.device);
  p.material = static_cast<const uint8_t*>(a->material.device);
  p.normal = static_cast<const uint8_t*>(a->normal.device);
  p.depth = static_cast<const uint8_t*>(a->depth.device);
  p.out = static_cast<uint8_t*>(a->out.device);
  p.lights = a->lights;
  p.pitch_albedo = a->albedo.pitch_bytes;
  p.pitch_material = a->material.pitch_bytes;
  p.pitch_normal = a->normal.pitch_bytes;
  p.pitch_depth = a->depth.pitch_bytes;
  p.pitch_out = a->out.pitch_bytes;
  p.width = w;
  p.height = h;
  p.row_offset = a->row_offset;
  p.full_height = full_h;
  p.light_count = a->light_count;
  p.flags = a->flags;
  switch (a->out.format) {
    case VKR_FORMAT_R16G16B16A16_SFLOAT: p.out_mode = kOutRgba16f; break;
    case VKR_FORMAT_R32G32B32A32_SFLOAT: p.out_mode = kOutRgba32f; break;
    case VKR_FORMAT_B8G8R8A8_UNORM: p.out_mode = kOutBgra8; break;
    default: p.out_mode = kOutRgba8; break;
  }
  p.cam[0] = a->push[0];
  p.cam[1] = a->push[1];
  p.cam[2] = a->push[2];
  p.radius2 = a->push[3] * a->push[3];
  std::memcpy(p.inv_vp, a->push + 4, 64);
  VKR_HIP(hipSetDevice(ctx->device));
  const TiledLayout layout = vkr_deferred_tiled_layout(p);
  if (layout.tiled) {
    if (stream_is_recording(ctx))
      return fail(VKR_ERR_BAD_STATE, "the radius-cutoff resolve keeps per-call state on the host and cannot be recorded into a HIP graph");
    if (!ctx->tile_counters) {
      VKR_HIP(hipMalloc(reinterpret_cast<void**>(&ctx->tile_counters), 256));
      VKR_HIP(hipMemsetAsync(ctx->tile_counters, 0, 256, ctx->stream));
    }
    if (!ctx->deep_hint && hipHostMalloc(reinterpret_cast<void**>(&ctx->deep_hint), sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
      *ctx->deep_hint = 0xFFFFFFFFu;  // (without the word every tiled call launches the deep-tile kernel: slower by a launch, not wrong)
      if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->deep_hint_device), ctx->deep_hint, 0) != hipSuccess) ctx->deep_hint_device = nullptr;
    }
    if (layout.total_bytes > ctx->tiled_ws_bytes) {
      // hipFree waits for the device: earlier launches still reading the old lists finish first
      if (ctx->tiled_ws) VKR_HIP(hipFree(ctx->tiled_ws));
      ctx->tiled_ws = nullptr;
      ctx->tiled_ws_bytes = 0;
      VKR_HIP(hipMalloc(&ctx->tiled_ws, layout.total_bytes));
      ctx->tiled_ws_bytes = layout.total_bytes;
    }
  }
  // the previous frame's number of deep tiles, if it has arrived (vkr_context::deep_hint); unknown counts as "some"
  const bool deep_expected = layout.tiled && (!ctx->deep_hint || *static_cast<volatile uint32_t*>(ctx->deep_hint) != 0u);
  const hipError_t launched = vkr_launch_deferred(p, ctx->tiled_ws, ctx->tile_counters, ctx->tile_epoch, (uint32_t)ctx->props.multiProcessorCount, deep_expected, ctx->deep_hint_device, ctx->stream);
  if (launched != hipSuccess) {
    if (layout.tiled) {
      // part of a tiled call may have been enqueued: the queue counters it used must not be taken for zero by the
      // next call, so both sets are cleared and the alternation starts over
      (void)hipMemsetAsync(ctx->tile_counters, 0, 256, ctx->stream);
      ctx->tile_epoch = 0;
    }
    return vkr_fail(launched == hipErrorOutOfMemory ? VKR_ERR_OUT_OF_MEMORY : VKR_ERR_DEVICE, "resolve launch: %s", hipGetErrorString(launched));
  }
  if (layout.tiled) {
    ctx->tile_epoch += 1u;
    ctx->last_tiled = layout;
    ctx->last_width = w;
    ctx->last_height = h;
    ctx->last_lights = a->light_count;
  }
  return VKR_OK;
}

// static_state: the allocation-free entry point runs the deterministic kernel in the library's per-GPU array.
int32_t run_cull(vkr_context* ctx, const vkr_cull_args* a, bool static_state = false) {
  if (!ctx || !a) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_prep_indirect: null argument");
  if (a->instance_count >= (1ull << 31)) return fail(VKR_ERR_INVALID_ARGUMENT, "instance_count %llu >= 2^31", (unsigned long long)a->instance_count);
  if (a->flags & ~(VKR_CULL_DETERMINISTIC | VKR_CULL_CMDS_PREZEROED)) return fail(VKR_ERR_INVALID_ARGUMENT, "unknown flags 0x%x", a->flags);
  static_assert(VKR_NOALLOC_MAX_INSTANCES == kCullStaticMaxInstances, "header limit = kernel limit");
  if (a->mesh_count == 0u) {
    if (a->instance_count != 0u) return fail(VKR_ERR_INVALID_ARGUMENT, "instances without meshes");
    return VKR_OK;
  }
  if (!a->mesh_base || !a->cmd_template || !a->cmd_out) return fail(VKR_ERR_INVALID_ARGUMENT, "mesh_base / cmd_template / cmd_out: null");
  if (a->instance_count != 0u && (!a->aabb_min || !a->aabb_max || !a->visible))
    return fail(VKR_ERR_INVALID_ARGUMENT, "aabb_min / aabb_max / visible: null with %llu instances", (unsigned long long)a->instance_count);
  if ((reinterpret_cast<uintptr_t>(a->aabb_min) | reinterpret_cast<uintptr_t>(a->aabb_max)) & 15u)
    return fail(VKR_ERR_INVALID_ARGUMENT, "aabb_min / aabb_max must be 16-byte aligned");
  CullParams p{};
  p.aabb_min = static_cast<const float4*>(a->aabb_min);
  p.aabb_max = static_cast<const float4*>(a->aabb_max);
  p.mesh_base = a->mesh_base;
  p.cmd_template = reinterpret_cast<const DrawCmd*>(a->cmd_template);
  p.cmd_out = reinterpret_cast<DrawCmd*>(a->cmd_out);
  p.visible = a->visible;
  p.error_flag = a->error_flag;
  p.instance_count = (uint32_t)a->instance_count;
  p.mesh_count = a->mesh_count;
  p.first_mesh = a->first_mesh;
  p.first_instance = a->first_instance;
  std::memcpy(p.planes, a->planes, sizeof(p.planes));
  VKR_HIP(hipSetDevice(ctx->device));
  const uint32_t resident = (uint32_t)ctx->props.multiProcessorCount * vkr_cull_blocks_per_cu();
  const uint64_t need = vkr_cull_state_bytes(a->instance_count);
  if (static_state) {
    const hipError_t e = vkr_launch_cull_static(p, ctx->device, resident, ctx->cull_fault_device, ctx->stream);
    if (e == hipErrorNotInitialized) return fail(VKR_ERR_BAD_STATE, "vkr_prep_indirect_noalloc: the static state of device %d was not resolved at context creation", ctx->device);
    if (e == hipErrorInvalidValue) return fail(VKR_ERR_INVALID_ARGUMENT, "static state too small for %llu instances", (unsigned long long)a->instance_count);
    VKR_HIP(e);
  } else if (a->flags & VKR_CULL_DETERMINISTIC) {
    if (stream_is_recording(ctx))
      return fail(VKR_ERR_BAD_STATE, "the deterministic cull keeps per-call state on the host and cannot be recorded into a HIP graph: use vkr_prep_indirect_noalloc");
    if (need > ctx->cull_state_bytes) {
      // hipFree waits for the device: an earlier launch still using the old state finishes first
      if (ctx->cull_state) VKR_HIP(hipFree(ctx->cull_state));
      ctx->cull_state = nullptr;
      ctx->cull_state_bytes = 0;
      VKR_HIP(hipMalloc(&ctx->cull_state, need));
      ctx->cull_state_bytes = need;
      ctx->cull_seq = 0;
    }
    if (ctx->cull_state && (ctx->cull_seq == 0u || ctx->cull_seq + 1u >= kCullSeqRecorded)) {  // fresh memory, or every sequence number has been used
      VKR_HIP(hipMemsetAsync(ctx->cull_state, 0, ctx->cull_state_bytes, ctx->stream));
      ctx->cull_seq = 0;
    }
    ctx->cull_seq += 1u;
    VKR_HIP(vkr_launch_cull_deterministic(p, ctx->cull_state, ctx->cull_seq, resident, ctx->cull_fault_device, ctx->stream));
  } else {
    VKR_HIP(vkr_launch_cull_atomic(p, (a->flags & VKR_CULL_CMDS_PREZEROED) != 0u, ctx->stream));
  }
  return VKR_OK;
}

int32_t image_from_binding(const BoundImage& b, const char* what, vkr_image2d* out) {
  if (!b.buf) return fail(VKR_ERR_BAD_STATE, "dispatch: %s image not bound", what);
  const uint64_t need = (uint64_t)b.pitch * b.height;
  if (b.offset > b.buf->bytes || need > b.buf->bytes - b.offset) return fail(VKR_ERR_INVALID_ARGUMENT, "dispatch: %s image exceeds its buffer", what);
  out->device = static_cast<uint8_t*>(b.buf->device) + b.offset;
  out->width = b.width;
  out->height = b.height;
  out->pitch_bytes = b.pitch;
  out->format = b.format;
  return VKR_OK;
}

int32_t buffer_ptr(const BoundBuffer& b, const char* what, uint64_t need, void** out) {
  if (!b.buf) return fail(VKR_ERR_BAD_STATE, "dispatch: %s buffer not bound", what);
  if (b.offset > b.buf->bytes || b.range > b.buf->bytes - b.offset || b.range < need)
    return fail(VKR_ERR_INVALID_ARGUMENT, "dispatch: %s buffer range %llu too small (need %llu)", what, (unsigned long long)b.range,
                (unsigned long long)need);
  *out = static_cast<uint8_t*>(b.buf->device) + b.offset;
  return VKR_OK;
}

int32_t replay(vkr_context* ctx, const Dispatch& d) {
  const BindState& s = d.state;
  if (!s.kernel) return fail(VKR_ERR_BAD_STATE, "dispatch without a bound kernel");
  if (s.push_bytes < s.kernel->push_min)
    return fail(VKR_ERR_BAD_STATE, "kernel %s needs %u bytes of push constants, %u were pushed", s.kernel->name, s.kernel->push_min, s.push_bytes);
  int32_t rc;
  if (s.kernel->id == kKernelDeferred) {
    vkr_deferred_args a{};
    if ((rc = image_from_binding(s.images[0][0][0], "albedo", &a.albedo)) != VKR_OK) return rc;
    if ((rc = image_from_binding(s.images[0][0][1], "material", &a.material)) != VKR_OK) return rc;
    if ((rc = image_from_binding(s.images[0][0][2], "normal", &a.normal)) != VKR_OK) return rc;
    if ((rc = image_from_binding(s.images[0][0][3], "depth", &a.depth)) != VKR_OK) return rc;
    if ((rc = image_from_binding(s.images[1][0][0], "output", &a.out)) != VKR_OK) return rc;
    const BoundBuffer& lb = s.buffers[0][1];
    void* lights = nullptr;
    if ((rc = buffer_ptr(lb, "light", 0, &lights)) != VKR_OK) return rc;
    if (lb.range % VKR_LIGHT_STRIDE) return fail(VKR_ERR_INVALID_ARGUMENT, "light buffer range %llu is not a multiple of 48", (unsigned long long)lb.range);
    a.lights = lights;
    a.light_count = (uint32_t)(lb.range / VKR_LIGHT_STRIDE);
    std::memcpy(a.push, s.push, 80);
    if (s.push_bytes >= 96) {
      uint32_t ext[4];
      std::memcpy(ext, s.push + 80, 16);
      a.flags = ext[0];
      a.row_offset = ext[1];
      a.full_height = ext[2];
    }
    if (d.x != a.out.width || d.y != a.out.height || d.z != 1u)
      return fail(VKR_ERR_INVALID_ARGUMENT, "deferred: dispatch(%u,%u,%u) must equal the output extent %ux%ux1", d.x, d.y, d.z, a.out.width, a.out.height);
    return run_deferred(ctx, &a);
  }
  // prep_indirect
  vkr_cull_args a{};
  uint32_t tail[4];
  std::memcpy(a.planes, s.push, 96);
  std::memcpy(tail, s.push + 96, 16);
  a.mesh_count = tail[0];
  a.first_mesh = tail[1];
  a.first_instance = tail[2];
  a.flags = tail[3];
  a.instance_count = d.x;
  if (d.y != 1u || d.z != 1u) return fail(VKR_ERR_INVALID_ARGUMENT, "prep_indirect: dispatch(%u,%u,%u) must be (N,1,1)", d.x, d.y, d.z);
  void* ptr = nullptr;
  const uint64_t n = a.instance_count, m = a.mesh_count;
  if ((rc = buffer_ptr(s.buffers[0][0], "aabb_min", n * 16, &ptr)) != VKR_OK) return rc;
  a.aabb_min = ptr;
  if ((rc = buffer_ptr(s.buffers[0][1], "aabb_max", n * 16, &ptr)) != VKR_OK) return rc;
  a.aabb_max = ptr;
  if ((rc = buffer_ptr(s.buffers[0][2], "mesh_base", m * 4, &ptr)) != VKR_OK) return rc;
  a.mesh_base = static_cast<const uint32_t*>(ptr);
  if ((rc = buffer_ptr(s.buffers[0][3], "cmd_template", m * 20, &ptr)) != VKR_OK) return rc;
  a.cmd_template = static_cast<const vkr_draw_indexed_indirect_command*>(ptr);
  if ((rc = buffer_ptr(s.buffers[0][4], "cmd_out", m * 20, &ptr)) != VKR_OK) return rc;
  a.cmd_out = static_cast<vkr_draw_indexed_indirect_command*>(ptr);
  if ((rc = buffer_ptr(s.buffers[0][5], "visible", n * 4, &ptr)) != VKR_OK) return rc;
  a.visible = static_cast<uint32_t*>(ptr);
  return run_cull(ctx, &a);
}

}  // namespace

int32_t vkr_fail(int32_t code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
  return code;
}

int32_t vkr_check_cull_fault(vkr_context* ctx) {
  if (!ctx || !ctx->cull_fault) return VKR_OK;
  const uint32_t seq = __atomic_exchange_n(ctx->cull_fault, 0u, __ATOMIC_RELAXED);
  if (seq == 0u) return VKR_OK;
  return vkr_fail(VKR_ERR_DEVICE, "a deterministic cull on this context (launch %u) stopped waiting for one of its own workgroups after %u s: "
                  "the device did not run them; visible[] and cmd_out[] of the deterministic culls since the last wait are invalid", seq, kCullPatienceSeconds);
}

extern "C" {

const char* vkr_last_error(void) { return g_last_error; }

int32_t vkr_context_create(int32_t device_ordinal, vkr_context** out_ctx) {
  if (!out_ctx) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_context_create: null out pointer");
  *out_ctx = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(VKR_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback", hipGetErrorString(e));
  if (device_ordinal < 0 || device_ordinal >= count) return fail(VKR_ERR_INVALID_ARGUMENT, "device ordinal %d out of range [0,%d)", device_ordinal, count);
  std::unique_ptr<vkr_context> ctx(new (std::nothrow) vkr_context);
  if (!ctx) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  ctx->device = device_ordinal;
  VKR_HIP(hipSetDevice(device_ordinal));
  VKR_HIP(hipGetDeviceProperties(&ctx->props, device_ordinal));
  if (std::strncmp(ctx->props.gcnArchName, "gfx950", 6) != 0)
    return fail(VKR_ERR_UNSUPPORTED, "device %d is %s; this library carries gfx950 code only", device_ordinal, ctx->props.gcnArchName);
  // resolves the per-GPU address of the static cull workspace now, so that vkr_prep_indirect_noalloc never
  // triggers a first-use code object load (which allocates) on this device
  VKR_HIP(vkr_cull_static_init(device_ordinal));
  (void)vkr_cull_blocks_per_cu();  // occupancy query of the deterministic cull, made here rather than inside a call
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->cull_fault), sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
      hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->cull_fault_device), ctx->cull_fault, 0) != hipSuccess) {
    if (ctx->cull_fault) (void)hipHostFree(ctx->cull_fault);
    return fail(VKR_ERR_OUT_OF_MEMORY, "no pinned host word for the context");
  }
  *ctx->cull_fault = 0u;
  const hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  if (se != hipSuccess) {
    (void)hipHostFree(ctx->cull_fault);
    return fail(VKR_ERR_DEVICE, "hipStreamCreateWithFlags: %s", hipGetErrorString(se));
  }
  *out_ctx = ctx.release();
  return VKR_OK;
}

int32_t vkr_context_destroy(vkr_context* ctx) {
  if (!ctx) return VKR_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamDestroy(ctx->stream);
  }
  if (ctx->cull_state) (void)hipFree(ctx->cull_state);
  if (ctx->tiled_ws) (void)hipFree(ctx->tiled_ws);
  if (ctx->tile_counters) (void)hipFree(ctx->tile_counters);
  if (ctx->deep_hint) (void)hipHostFree(ctx->deep_hint);
  if (ctx->cull_fault) (void)hipHostFree(ctx->cull_fault);
  delete ctx;
  return VKR_OK;
}

int32_t vkr_context_device_info(vkr_context* ctx, char* out, size_t out_bytes, int32_t* out_cu_count, int32_t* out_clock_mhz) {
  if (!ctx) return fail(VKR_ERR_INVALID_ARGUMENT, "null context");
  if (out && out_bytes)
    snprintf(out, out_bytes, "%s %s CUs=%d clock=%dMHz", ctx->props.name, ctx->props.gcnArchName, ctx->props.multiProcessorCount,
             ctx->props.clockRate / 1000);
  if (out_cu_count) *out_cu_count = ctx->props.multiProcessorCount;
  if (out_clock_mhz) *out_clock_mhz = ctx->props.clockRate / 1000;
  return VKR_OK;
}

int32_t vkr_context_self_test(vkr_context* ctx, uint32_t samples, uint32_t* out_failures) {
  if (!ctx || !out_failures) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  VKR_HIP(hipSetDevice(ctx->device));
  uint32_t* d = nullptr;
  VKR_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof(uint32_t)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(uint32_t), ctx->stream);
  if (e == hipSuccess && samples) e = vkr_launch_exact_div_check(samples, 0x9E3779B9u, d, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(out_failures, d, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(VKR_ERR_DEVICE, "self test: %s", hipGetErrorString(e));
  return VKR_OK;
}

int32_t vkr_context_stream(vkr_context* ctx, void** out_hip_stream) {
  if (!ctx || !out_hip_stream) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  *out_hip_stream = ctx->stream;
  return VKR_OK;
}

int32_t vkr_buffer_create(vkr_context* ctx, uint64_t bytes, uint32_t flags, vkr_buffer** out_buf) {
  if (!ctx || !out_buf) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_buffer_create: null argument");
  *out_buf = nullptr;
  if (flags > VKR_BUFFER_HOST_VISIBLE) return fail(VKR_ERR_INVALID_ARGUMENT, "unknown buffer flags 0x%x", flags);
  std::unique_ptr<vkr_buffer> b(new (std::nothrow) vkr_buffer);
  if (!b) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  b->ctx = ctx;
  b->bytes = bytes;
  VKR_HIP(hipSetDevice(ctx->device));
  const uint64_t alloc = bytes ? bytes : 16;
  if (flags == VKR_BUFFER_HOST_VISIBLE) {
    VKR_HIP(hipHostMalloc(&b->host, alloc, hipHostMallocMapped));
    VKR_HIP(hipHostGetDevicePointer(&b->device, b->host, 0));
  } else {
    VKR_HIP(hipMalloc(&b->device, alloc));
  }
  *out_buf = b.release();
  return VKR_OK;
}

int32_t vkr_buffer_wrap(vkr_context* ctx, void* device_ptr, uint64_t bytes, vkr_buffer** out_buf) {
  if (!ctx || !out_buf || (!device_ptr && bytes)) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_buffer_wrap: null argument");
  vkr_buffer* b = new (std::nothrow) vkr_buffer;
  if (!b) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  b->ctx = ctx;
  b->device = device_ptr;
  b->bytes = bytes;
  b->owned = false;
  *out_buf = b;
  return VKR_OK;
}

int32_t vkr_buffer_destroy(vkr_buffer* buf) {
  if (!buf) return VKR_OK;
  if (buf->owned) {
    (void)hipSetDevice(buf->ctx->device);
    if (buf->host)
      (void)hipHostFree(buf->host);
    else if (buf->device)
      (void)hipFree(buf->device);
  }
  delete buf;
  return VKR_OK;
}

int32_t vkr_buffer_size(vkr_buffer* buf, uint64_t* out_bytes) {
  if (!buf || !out_bytes) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  *out_bytes = buf->bytes;
  return VKR_OK;
}

int32_t vkr_buffer_map(vkr_buffer* buf, void** out_host_ptr) {
  if (!buf || !out_host_ptr) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (!buf->host) return fail(VKR_ERR_INVALID_ARGUMENT, "buffer is not HOST_VISIBLE");
  *out_host_ptr = buf->host;
  return VKR_OK;
}

int32_t vkr_buffer_device_ptr(vkr_buffer* buf, void** out_device_ptr) {
  if (!buf || !out_device_ptr) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  *out_device_ptr = buf->device;
  return VKR_OK;
}

int32_t vkr_buffer_upload(vkr_buffer* buf, uint64_t offset, const void* src, uint64_t bytes) {
  if (!buf || (!src && bytes)) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(VKR_ERR_INVALID_ARGUMENT, "upload of %llu bytes at %llu exceeds buffer size %llu", (unsigned long long)bytes, (unsigned long long)offset, (unsigned long long)buf->bytes);
  if (!bytes) return VKR_OK;
  VKR_HIP(hipSetDevice(buf->ctx->device));
  VKR_HIP(hipMemcpyAsync(static_cast<uint8_t*>(buf->device) + offset, src, bytes, hipMemcpyHostToDevice, buf->ctx->stream));
  return VKR_OK;
}

int32_t vkr_buffer_download(vkr_buffer* buf, uint64_t offset, void* dst, uint64_t bytes) {
  if (!buf || (!dst && bytes)) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(VKR_ERR_INVALID_ARGUMENT, "download of %llu bytes at %llu exceeds buffer size %llu", (unsigned long long)bytes, (unsigned long long)offset, (unsigned long long)buf->bytes);
  if (!bytes) return VKR_OK;
  VKR_HIP(hipSetDevice(buf->ctx->device));
  VKR_HIP(hipMemcpyAsync(dst, static_cast<uint8_t*>(buf->device) + offset, bytes, hipMemcpyDeviceToHost, buf->ctx->stream));
  VKR_HIP(hipStreamSynchronize(buf->ctx->stream));
  return vkr_check_cull_fault(buf->ctx);
}

int32_t vkr_buffer_fill_u32(vkr_buffer* buf, uint64_t offset, uint32_t value, uint64_t bytes) {
  if (!buf) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if ((offset | bytes) & 3u) return fail(VKR_ERR_INVALID_ARGUMENT, "fill offset/size must be multiples of 4");
  if (offset > buf->bytes || bytes > buf->bytes - offset) return fail(VKR_ERR_INVALID_ARGUMENT, "fill exceeds buffer size");
  if (!bytes) return VKR_OK;
  VKR_HIP(hipSetDevice(buf->ctx->device));
  VKR_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(static_cast<uint8_t*>(buf->device) + offset), (int)value, bytes / 4, buf->ctx->stream));
  return VKR_OK;
}

int32_t vkr_deferred_resolve(vkr_context* ctx, const vkr_deferred_args* args) { return run_deferred(ctx, args); }

int32_t vkr_deferred_tile_stats(vkr_context* ctx, vkr_tile_stats* out) {
  if (!ctx || !out) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_deferred_tile_stats: null argument");
  std::memset(out, 0, sizeof(*out));
  const TiledLayout& t = ctx->last_tiled;
  if (!t.tiled || !ctx->tiled_ws) return fail(VKR_ERR_BAD_STATE, "no resolve with light lists has run on this context");
  const uint64_t tiles = (uint64_t)t.tiles_x * t.tiles_y;
  std::unique_ptr<uint32_t[]> counts(new (std::nothrow) uint32_t[tiles]);
  std::unique_ptr<uint32_t[]> bin_counts(t.bins ? new (std::nothrow) uint32_t[t.bins] : nullptr);
  if (!counts || (t.bins && !bin_counts)) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  VKR_HIP(hipSetDevice(ctx->device));
  const uint8_t* ws = static_cast<const uint8_t*>(ctx->tiled_ws);
  VKR_HIP(hipMemcpyAsync(counts.get(), ws + t.off_tile_count, tiles * 4u, hipMemcpyDeviceToHost, ctx->stream));
  if (t.bins) VKR_HIP(hipMemcpyAsync(bin_counts.get(), ws + t.off_bin_count, t.bins * 4u, hipMemcpyDeviceToHost, ctx->stream));
  std::unique_ptr<uint16_t[]> kept(new (std::nothrow) uint16_t[tiles * 8u]);  // per group of a deep tile: lights it walked
  if (!kept) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  VKR_HIP(hipMemcpyAsync(kept.get(), ws + t.off_group_count, tiles * 16u, hipMemcpyDeviceToHost, ctx->stream));
  VKR_HIP(hipStreamSynchronize(ctx->stream));
  out->tiles = (uint32_t)tiles;
  out->list_cap = t.list_cap;
  out->pixels = (uint64_t)ctx->last_width * ctx->last_height;
  for (uint32_t ty = 0; ty < t.tiles_y; ++ty) {
    for (uint32_t tx = 0; tx < t.tiles_x; ++tx) {
      const uint32_t word = counts[(uint64_t)ty * t.tiles_x + tx];
      const uint64_t px = (uint64_t)std::min(32u, ctx->last_width - tx * 32u) * std::min(32u, ctx->last_height - ty * 32u);
      uint64_t walked = word & kTileCountMask;
      if (word == 0u) {
        out->dark_tiles += 1u;
      } else if (word & kTileOverflow) {
        out->overflow_tiles += 1u;
        walked = t.bins ? bin_counts[(uint64_t)(ty * 32u / kBinPixels) * t.bins_x + tx * 32u / kBinPixels] : ctx->last_lights;
      } else {
        out->listed_tiles += 1u;
        if (walked > out->max_list) out->max_list = (uint32_t)walked;
        if (word & kTileDeep) {  // its groups of 128 pixels (the last one may be smaller) walked their own narrowed lists
          out->deep_tiles += 1u;
          walked = 0;
          for (uint64_t g = 0; g * 128u < px; ++g) {
            const uint64_t own = kept[((uint64_t)ty * t.tiles_x + tx) * 8u + g];
            out->light_pixels += (own == kGroupOverflow ? (uint64_t)(word & kTileCountMask) : own) * std::min<uint64_t>(128u, px - g * 128u);
          }
        }
      }
      out->light_pixels += walked * px;
    }
  }
  return VKR_OK;
}
int32_t vkr_prep_indirect(vkr_context* ctx, const vkr_cull_args* args) { return run_cull(ctx, args); }
int32_t vkr_prep_indirect_reset(vkr_context* ctx, const vkr_cull_args* a) {
  if (!ctx || !a) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_prep_indirect_reset: null argument");
  if (a->mesh_count == 0u) return VKR_OK;
  if (!a->mesh_base || !a->cmd_template || !a->cmd_out) return fail(VKR_ERR_INVALID_ARGUMENT, "mesh_base / cmd_template / cmd_out: null");
  CullParams p{};
  p.mesh_base = a->mesh_base;
  p.cmd_template = reinterpret_cast<const DrawCmd*>(a->cmd_template);
  p.cmd_out = reinterpret_cast<DrawCmd*>(a->cmd_out);
  p.mesh_count = a->mesh_count;
  p.first_instance = a->first_instance;
  VKR_HIP(hipSetDevice(ctx->device));
  VKR_HIP(vkr_launch_cull_reset(p, ctx->stream));
  return VKR_OK;
}
int32_t vkr_prep_indirect_noalloc(vkr_context* ctx, const vkr_cull_args* args) {
  if (!ctx || !args) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_prep_indirect_noalloc: null argument");
  if (args->instance_count > VKR_NOALLOC_MAX_INSTANCES)
    return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_prep_indirect_noalloc: %llu instances exceed the limit of %llu (use vkr_prep_indirect)",
                (unsigned long long)args->instance_count, (unsigned long long)VKR_NOALLOC_MAX_INSTANCES);
  return run_cull(ctx, args, true);
}

int32_t vkr_kernel_get(vkr_context* ctx, const char* name, vkr_kernel** out_kernel) {
  if (!ctx || !name || !out_kernel) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_kernel_get: null argument");
  for (auto& k : g_kernels) {
    if (std::strcmp(k.name, name) == 0) {
      *out_kernel = &k;
      return VKR_OK;
    }
  }
  *out_kernel = nullptr;
  return fail(VKR_ERR_NOT_FOUND, "no kernel named '%s' (have: deferred, prep_indirect)", name);
}

int32_t vkr_kernel_push_constant_size(vkr_kernel* kernel, uint32_t* out_min_bytes, uint32_t* out_max_bytes) {
  if (!kernel) return fail(VKR_ERR_INVALID_ARGUMENT, "null kernel");
  if (out_min_bytes) *out_min_bytes = kernel->push_min;
  if (out_max_bytes) *out_max_bytes = kernel->push_max;
  return VKR_OK;
}

int32_t vkr_cmd_begin(vkr_context* ctx, vkr_cmd** out_cmd) {
  if (!ctx || !out_cmd) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_cmd_begin: null argument");
  vkr_cmd* c = new (std::nothrow) vkr_cmd;
  if (!c) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  c->ctx = ctx;
  *out_cmd = c;
  return VKR_OK;
}

int32_t vkr_cmd_bind_kernel(vkr_cmd* cmd, vkr_kernel* kernel) {
  if (!cmd || !kernel) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  cmd->cur.kernel = kernel;
  return VKR_OK;
}

int32_t vkr_cmd_push_constants(vkr_cmd* cmd, uint32_t offset, uint32_t size, const void* data) {
  if (!cmd || (!data && size)) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  if ((uint64_t)offset + size > sizeof(cmd->cur.push)) return fail(VKR_ERR_INVALID_ARGUMENT, "push constants [%u,%u) exceed 128 bytes", offset, offset + size);
  if ((offset | size) & 3u) return fail(VKR_ERR_INVALID_ARGUMENT, "push constant offset/size must be multiples of 4");
  std::memcpy(cmd->cur.push + offset, data, size);
  if (offset + size > cmd->cur.push_bytes) cmd->cur.push_bytes = offset + size;
  return VKR_OK;
}

int32_t vkr_cmd_bind_buffer(vkr_cmd* cmd, uint32_t set, uint32_t binding, vkr_buffer* buf, uint64_t offset, uint64_t range) {
  if (!cmd || !buf) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  if (set >= kMaxSets || binding >= kMaxBindings) return fail(VKR_ERR_INVALID_ARGUMENT, "set %u / binding %u out of range", set, binding);
  if (offset > buf->bytes || range > buf->bytes - offset) return fail(VKR_ERR_INVALID_ARGUMENT, "buffer binding [%llu,%llu) exceeds buffer size %llu", (unsigned long long)offset, (unsigned long long)(offset + range), (unsigned long long)buf->bytes);
  cmd->cur.buffers[set][binding] = {buf, offset, range};
  return VKR_OK;
}

int32_t vkr_cmd_bind_image2d(vkr_cmd* cmd, uint32_t set, uint32_t binding, uint32_t array_element, vkr_buffer* buf, uint64_t offset,
                             uint32_t width, uint32_t height, uint32_t pitch_bytes, uint32_t format) {
  if (!cmd || !buf) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  if (set >= kMaxSets || binding >= kMaxBindings || array_element >= kMaxArray)
    return fail(VKR_ERR_INVALID_ARGUMENT, "set %u / binding %u / element %u out of range", set, binding, array_element);
  if (format_texel_bytes(format) == 0) return fail(VKR_ERR_INVALID_ARGUMENT, "unsupported image format %u", format);
  cmd->cur.images[set][binding][array_element] = {buf, offset, width, height, pitch_bytes, format};
  return VKR_OK;
}

int32_t vkr_cmd_dispatch(vkr_cmd* cmd, uint32_t x, uint32_t y, uint32_t z) {
  if (!cmd) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  if (!cmd->cur.kernel) return fail(VKR_ERR_BAD_STATE, "dispatch without a bound kernel");
  try {
    cmd->dispatches.push_back(Dispatch{cmd->cur, x, y, z});
  } catch (const std::bad_alloc&) {
    return fail(VKR_ERR_OUT_OF_MEMORY, "vkr_cmd_dispatch: out of host memory");
  }
  return VKR_OK;
}

int32_t vkr_cmd_end(vkr_cmd* cmd) {
  if (!cmd) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  if (cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list already ended");
  cmd->ended = true;
  return VKR_OK;
}

int32_t vkr_cmd_destroy(vkr_cmd* cmd) {
  delete cmd;
  return VKR_OK;
}

static int32_t make_fence(vkr_context* ctx, vkr_fence** out_fence) {
  vkr_fence* f = new (std::nothrow) vkr_fence;
  if (!f) return fail(VKR_ERR_OUT_OF_MEMORY, "out of host memory");
  f->ctx = ctx;
  hipError_t e = hipEventCreate(&f->event);
  if (e == hipSuccess) e = hipEventRecord(f->event, ctx->stream);
  if (e != hipSuccess) {
    if (f->event) (void)hipEventDestroy(f->event);
    delete f;
    return fail(VKR_ERR_DEVICE, "fence creation failed: %s", hipGetErrorString(e));
  }
  *out_fence = f;
  return VKR_OK;
}

int32_t vkr_queue_submit(vkr_context* ctx, vkr_cmd* cmd, vkr_fence** out_fence) {
  if (!ctx || !cmd) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_queue_submit: null argument");
  if (cmd->ctx != ctx) return fail(VKR_ERR_INVALID_ARGUMENT, "command list belongs to another context");
  if (!cmd->ended) return fail(VKR_ERR_BAD_STATE, "command list not ended");
  if (out_fence) *out_fence = nullptr;
  VKR_HIP(hipSetDevice(ctx->device));
  for (const Dispatch& d : cmd->dispatches) {
    const int32_t rc = replay(ctx, d);
    if (rc != VKR_OK) return rc;
  }
  if (out_fence) return make_fence(ctx, out_fence);
  return VKR_OK;
}

int32_t vkr_queue_marker(vkr_context* ctx, vkr_fence** out_fence) {
  if (!ctx || !out_fence) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  VKR_HIP(hipSetDevice(ctx->device));
  return make_fence(ctx, out_fence);
}

int32_t vkr_queue_wait_idle(vkr_context* ctx) {
  if (!ctx) return fail(VKR_ERR_INVALID_ARGUMENT, "null context");
  VKR_HIP(hipSetDevice(ctx->device));
  VKR_HIP(hipStreamSynchronize(ctx->stream));
  return vkr_check_cull_fault(ctx);
}

int32_t vkr_fence_status(vkr_fence* fence) {
  if (!fence) return fail(VKR_ERR_INVALID_ARGUMENT, "null fence");
  const hipError_t e = hipEventQuery(fence->event);
  if (e == hipSuccess) return vkr_check_cull_fault(fence->ctx);
  if (e == hipErrorNotReady) return VKR_ERR_NOT_READY;
  return fail(VKR_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(e));
}

int32_t vkr_fence_wait(vkr_fence* fence, uint64_t timeout_ns) {
  if (!fence) return fail(VKR_ERR_INVALID_ARGUMENT, "null fence");
  if (timeout_ns == UINT64_MAX) {
    VKR_HIP(hipEventSynchronize(fence->event));
    return vkr_check_cull_fault(fence->ctx);
  }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::nanoseconds(timeout_ns);
  for (;;) {
    const hipError_t e = hipEventQuery(fence->event);
    if (e == hipSuccess) return vkr_check_cull_fault(fence->ctx);
    if (e != hipErrorNotReady) return fail(VKR_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(e));
    if (std::chrono::steady_clock::now() >= deadline) return VKR_ERR_NOT_READY;
    std::this_thread::yield();
  }
}

int32_t vkr_fence_elapsed_ms(vkr_fence* begin, vkr_fence* end, float* out_ms) {
  if (!begin || !end || !out_ms) return fail(VKR_ERR_INVALID_ARGUMENT, "null argument");
  VKR_HIP(hipEventElapsedTime(out_ms, begin->event, end->event));
  return VKR_OK;
}

int32_t vkr_fence_destroy(vkr_fence* fence) {
  if (!fence) return VKR_OK;
  if (fence->event) (void)hipEventDestroy(fence->event);
  delete fence;
  return VKR_OK;
}

void vkr_perspective(float fovy, float aspect, float z_near, float z_far, float out_m[16]) { vkrm::perspective(fovy, aspect, z_near, z_far, out_m); }
void vkr_look_at(const float eye[3], const float center[3], const float up[3], float out_m[16]) { vkrm::look_at(eye, center, up, out_m); }
void vkr_mat4_mul(const float a[16], const float b[16], float out_m[16]) { vkrm::mul(a, b, out_m); }
int32_t vkr_mat4_inverse(const float m[16], float out_m[16]) {
  if (!vkrm::inverse(m, out_m)) return fail(VKR_ERR_INVALID_ARGUMENT, "vkr_mat4_inverse: singular matrix");
  return VKR_OK;
}
void vkr_frustum_planes(const float vp[16], float out_planes[24]) { vkrm::frustum_planes(vp, out_planes); }

}  // extern "C"
