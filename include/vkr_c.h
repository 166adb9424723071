/*
 * vkr_c.h - C ABI of the MI355X (gfx950, HIP) implementation of vk-renderer's two
 * compute passes: the deferred lighting resolve and the frustum cull / indirect
 * command compaction.
 *
 * Every function returns a vkr_status (0 = OK).  No C++ exceptions cross this ABI;
 * vkr_last_error() returns the message of the calling thread's last failure.
 * All pointers named `device` are HIP device pointers (hipMalloc'd, or any memory
 * the GPU can address); they may come from vkr_buffer_* or from the caller's own
 * allocator (e.g. a torch tensor's data_ptr()).
 *
 * A context owns one HIP stream on one GPU and is NOT thread-safe, exactly like
 * the reference's single-threaded wrappers (VMA is created externally
 * synchronised at /root/reference/src/vkmemory.cpp:176; one command pool per
 * queue without locks at /root/reference/inc/vkcore.hpp:124-168).
 *
 * Each entry point names the reference interface it replaces.
 */
#ifndef VKR_C_H_
#define VKR_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VKR_API __attribute__((visibility("default")))

typedef enum vkr_status {
  VKR_OK = 0,
  VKR_ERR_INVALID_ARGUMENT = 1,
  VKR_ERR_OUT_OF_MEMORY = 2,
  VKR_ERR_DEVICE = 3,      /* a HIP call failed; see vkr_last_error() */
  VKR_ERR_NOT_FOUND = 4,   /* unknown kernel name */
  VKR_ERR_NOT_READY = 5,   /* fence not signalled yet (status query / timed-out wait) */
  VKR_ERR_BAD_STATE = 6,   /* command list used out of order */
  VKR_ERR_UNSUPPORTED = 7
} vkr_status;

/* Image formats use the numeric values of VkFormat so that host code written against
 * the reference (vk::Format at /root/reference/src/main.cpp:112-117) ports unchanged. */
typedef enum vkr_format {
  VKR_FORMAT_UNDEFINED = 0,
  VKR_FORMAT_R8G8B8A8_UNORM = 37,
  VKR_FORMAT_B8G8R8A8_UNORM = 44, /* output image only: the swapchain format the reference binds
                                     (/root/reference/src/vkpresent.cpp:17); bytes in memory are B, G, R, A */
  VKR_FORMAT_R16G16B16A16_SFLOAT = 97,
  VKR_FORMAT_R32_SFLOAT = 100,
  VKR_FORMAT_R32G32B32A32_SFLOAT = 109,
  VKR_FORMAT_D32_SFLOAT = 126
} vkr_format;

typedef struct vkr_context vkr_context;
typedef struct vkr_buffer vkr_buffer;
typedef struct vkr_kernel vkr_kernel;
typedef struct vkr_cmd vkr_cmd;
typedef struct vkr_fence vkr_fence;

/* ------------------------------------------------------------------ context / errors */

/* Replaces vw::Instance + vw::Device bring-up (/root/reference/inc/vkcore.hpp:186-218) and the
 * process-global vw::g::device (/root/reference/src/vkcore.cpp:4-10): one context per GPU. */
VKR_API int32_t vkr_context_create(int32_t device_ordinal, vkr_context** out_ctx);
VKR_API int32_t vkr_context_destroy(vkr_context* ctx);
/* Message of the calling thread's most recent failing call ("" if none). */
VKR_API const char* vkr_last_error(void);
/* Fills a NUL-terminated description "<name> <gcn arch> CUs=<n> clock=<MHz>". */
VKR_API int32_t vkr_context_device_info(vkr_context* ctx, char* out, size_t out_bytes, int32_t* out_cu_count,
                                         int32_t* out_clock_mhz);
/* Runs the library's device self-checks (currently: the resolve's correctly-rounded division against
 * IEEE division on `samples` operand pairs) and returns the number of failures in *out_failures. */
VKR_API int32_t vkr_context_self_test(vkr_context* ctx, uint32_t samples, uint32_t* out_failures);
/* The HIP stream (hipStream_t) all work of this context is ordered on. */
VKR_API int32_t vkr_context_stream(vkr_context* ctx, void** out_hip_stream);

/* ------------------------------------------------------------------ memory
 * Replaces vw::MemoryAllocator / vw::Buffer / vw::Image / vw::StagingBuffer
 * (/root/reference/inc/vkmemory.hpp:25-219): VMA allocations become hipMalloc (device) or pinned
 * host memory mapped into the device (the CPU_TO_GPU persistently-mapped case,
 * /root/reference/src/vkmemory.cpp:34-45). */
#define VKR_BUFFER_DEVICE 0u       /* VMA_MEMORY_USAGE_GPU_ONLY */
#define VKR_BUFFER_HOST_VISIBLE 1u /* VMA_MEMORY_USAGE_CPU_TO_GPU: persistently mapped */

VKR_API int32_t vkr_buffer_create(vkr_context* ctx, uint64_t bytes, uint32_t flags, vkr_buffer** out_buf);
/* Wraps memory the caller owns; destroy only drops the handle. */
VKR_API int32_t vkr_buffer_wrap(vkr_context* ctx, void* device_ptr, uint64_t bytes, vkr_buffer** out_buf);
VKR_API int32_t vkr_buffer_destroy(vkr_buffer* buf);
VKR_API int32_t vkr_buffer_size(vkr_buffer* buf, uint64_t* out_bytes);
/* Host pointer of a HOST_VISIBLE buffer (vw::Buffer::copyToMapped's mMappedPtr). */
VKR_API int32_t vkr_buffer_map(vkr_buffer* buf, void** out_host_ptr);
VKR_API int32_t vkr_buffer_device_ptr(vkr_buffer* buf, void** out_device_ptr);
/* Stream-ordered copies (vw::StagingBuffer::queueBufferCopy + flush); the host range must stay
 * valid until the next fence wait / wait_idle. */
VKR_API int32_t vkr_buffer_upload(vkr_buffer* buf, uint64_t offset, const void* src, uint64_t bytes);
VKR_API int32_t vkr_buffer_download(vkr_buffer* buf, uint64_t offset, void* dst, uint64_t bytes);
VKR_API int32_t vkr_buffer_fill_u32(vkr_buffer* buf, uint64_t offset, uint32_t value, uint64_t bytes);

/* ------------------------------------------------------------------ typed entry points */

typedef struct vkr_image2d {
  void* device;         /* first texel of row 0 */
  uint32_t width;       /* texels */
  uint32_t height;      /* rows */
  uint32_t pitch_bytes; /* >= 16-byte-aligned row size; rows must start 16-byte aligned */
  uint32_t format;      /* vkr_format */
} vkr_image2d;

#define VKR_LIGHT_STRIDE 48u /* std140 Light record, /root/reference/shaders/deferred.comp:12-19 */

#define VKR_DEFERRED_RADIUS_CUTOFF 1u /* extension: a light contributes only where dist^2 < pointLightRadius^2
                                         (push[3]); enables per-tile light lists.  Off = reference behaviour
                                         (pointLightRadius is pushed but never read, deferred.comp:8,120-129).
                                         The call lists the lights that can reach each 32 x 32 pixel tile (up to 65536
                                         lights; from 256 lights up via a coarse level of 128 x 128 pixel bins), writes
                                         the tiles no light reaches straight away (their pixels are their ambient term)
                                         and shades the others from their lists; a tile whose pixels span more depth
                                         than the radius is shaded in eight groups of pixels of similar depth, each from
                                         the part of the list that can reach it.  Lists, bins, the queues of lit tiles,
                                         the depth order and group lists live in context-owned device memory (5 KiB +
                                         40 bytes per tile, 4 bytes per bin and light; allocated or grown inside the
                                         call, released with the context).
                                         VKR_ERR_BAD_STATE while the context's stream is being captured into a HIP
                                         graph: the launches take a value the host advances per call. */
#define VKR_DEFERRED_NO_TILE_LISTS 2u /* with RADIUS_CUTOFF: apply the cutoff per pixel only and loop over the whole
                                         light table (diagnostic; results are bit-identical to the tiled path) */
#define VKR_DEFERRED_WHOLE_TILE_WORKGROUPS 4u /* whole-table resolve: always the loop that evaluates every light on
                                         every pixel, one 256-thread workgroup per 32 x 32 pixel tile.  Without it a
                                         table of 24 lights or more is resolved from per-pixel lists (every pixel walks only
                                         the lights that are not behind its surface: the ones deferred.comp:43,84 does
                                         not return 0 for), and a shorter table in a launch of fewer than four rounds of
                                         such workgroups (a small frame, a screen band) by single-wave workgroups of 64
                                         pixels of a row (diagnostic; bit-identical results either way) */

/* One dispatch of /root/reference/shaders/deferred.comp (main, :98-131) as recorded at
 * /root/reference/src/main.cpp:222-226.  Bindings follow main.cpp:163-167,190-196:
 * samplers[0..3] = albedo, material ("gSpecular"), normal, depth; UBO = lights; outTex = out. */
typedef struct vkr_deferred_args {
  vkr_image2d albedo;   /* R8G8B8A8_UNORM   (.rgb) */
  vkr_image2d material; /* R8G8B8A8_UNORM   (.g roughness, .b metalness) */
  vkr_image2d normal;   /* R16G16B16A16_SFLOAT (.rgb, un-normalised) */
  vkr_image2d depth;    /* D32_SFLOAT / R32_SFLOAT */
  vkr_image2d out;      /* R8G8B8A8_UNORM / B8G8R8A8_UNORM (reference), or R16G16B16A16_SFLOAT / R32G32B32A32_SFLOAT
                           (pre-quantisation colour, alpha = 1) */
  const void* lights;   /* device: light_count records of VKR_LIGHT_STRIDE bytes */
  uint32_t light_count;
  float push[20];       /* DeferredPushData, 80 bytes (/root/reference/src/main.cpp:76-80):
                           cameraPos @0, pointLightRadius @12, inverseVP @16 column-major */
  /* Screen-band sharding: the five images hold rows [row_offset, row_offset+height) of a frame
   * full_height rows tall (UV is computed against the full frame).  0/0 = whole frame. */
  uint32_t row_offset;
  uint32_t full_height;
  uint32_t flags;
  uint32_t reserved;
} vkr_deferred_args;

VKR_API int32_t vkr_deferred_resolve(vkr_context* ctx, const vkr_deferred_args* args);

/* Measurement support, no counterpart in the reference: what the light lists of the context's most recent
 * resolve with VKR_DEFERRED_RADIUS_CUTOFF looked like.  light_pixels / pixels is the mean number of lights a
 * pixel's loop walked (the L that the work of the frame is proportional to).  Waits for the context's stream. */
typedef struct vkr_tile_stats {
  uint32_t tiles;          /* 32 x 32 pixel tiles of the frame */
  uint32_t dark_tiles;     /* no light reaches them: written as ambient colour by the listing launch */
  uint32_t listed_tiles;   /* shaded from their own list */
  uint32_t overflow_tiles; /* more reaching lights than a list holds: shaded from their bin's candidates / the table */
  uint32_t max_list;       /* longest list among the listed tiles */
  uint32_t list_cap;       /* entries a list holds */
  uint64_t light_pixels;   /* sum over tiles (for deep tiles: over their groups of pixels) of lights walked x pixels inside the image */
  uint64_t pixels;         /* width x height */
  uint32_t deep_tiles;     /* listed tiles that span more depth than a light reaches: shaded in eight groups of pixels
                              of similar depth, each group walking the part of the tile's list that can reach it */
  uint32_t reserved;
} vkr_tile_stats;
VKR_API int32_t vkr_deferred_tile_stats(vkr_context* ctx, vkr_tile_stats* out);

/* VkDrawIndexedIndirectCommand, 20 bytes, one per mesh
 * (/root/reference/src/vkmodel.cpp:79,96; consumed at /root/reference/inc/vkmodel.hpp:72). */
typedef struct vkr_draw_indexed_indirect_command {
  uint32_t indexCount;
  uint32_t instanceCount;
  uint32_t firstIndex;
  int32_t vertexOffset;
  uint32_t firstInstance;
} vkr_draw_indexed_indirect_command;

#define VKR_CULL_DETERMINISTIC 1u /* visible[] ascending inside each mesh segment (one launch; its state, one 64-bit word
                                     per 1024 instances, lives in context-owned device memory that is allocated or grown
                                     inside the call and released with the context; VKR_ERR_BAD_STATE while the
                                     context's stream is being captured into a HIP graph - vkr_prep_indirect_noalloc
                                     is the form that records); default is one atomic per wave and mesh run (segment
                                     order unspecified) */

#define VKR_CULL_CMDS_PREZEROED 2u /* atomic mode only (ignored with VKR_CULL_DETERMINISTIC, whose launch writes every
                                     command whole): cmd_out[] already holds what vkr_prep_indirect_reset leaves there,
                                     so the call skips the launch that would write it (4.6 us of GPU time at 1024
                                     meshes; calls issued back to back take as long either way, 27.5 us at 2^22
                                     instances, because that launch runs in the shadow of the previous call's tail).  A
                                     frame loop resets the commands once its draws have consumed them - behind the
                                     frame, off the path from camera to commands. */

/* The pass /root/reference/shaders/prep_indirect.comp:1-5 leaves empty.  Instance i carries a
 * world-space AABB and a mesh id; instances of a mesh are contiguous, as the model-matrix array of
 * /root/reference/src/vkmodel.cpp:108-110 is (PerMeshData::modelMatrixBaseIndex,
 * /root/reference/inc/vkmodel.hpp:76-78).  An instance is visible iff for all six planes
 * ((n.x*p.x + n.y*p.y) + n.z*p.z) + n.w >= 0 at the AABB's p-vertex, in binary32 without fused
 * multiply-add.  Results: cmd_out[m] = cmd_template[m] with instanceCount = visible instances of
 * mesh m and firstInstance = first_instance + mesh_base[m]; visible[mesh_base[m] + k] = global
 * index (first_instance + i) of the k-th visible instance of mesh m. */
typedef struct vkr_cull_args {
  const void* aabb_min;     /* device: N x float4, xyz = min corner, w = mesh id (uint32 bits) */
  const void* aabb_max;     /* device: N x float4, xyz = max corner, w ignored */
  uint64_t instance_count;  /* N (< 2^31) */
  float planes[24];         /* six float4, see vkr_frustum_planes */
  const uint32_t* mesh_base;                               /* device: M entries, local to this call */
  uint32_t mesh_count;                                     /* M */
  uint32_t first_mesh;      /* instance-range sharding: mesh ids in aabb_min.w are global, this call owns
                               meshes [first_mesh, first_mesh + M) */
  uint32_t first_instance;  /* global index of local instance 0 */
  uint32_t flags;
  const vkr_draw_indexed_indirect_command* cmd_template;   /* device: M */
  vkr_draw_indexed_indirect_command* cmd_out;              /* device: M */
  uint32_t* visible;        /* device: N */
  uint32_t* error_flag;     /* device, optional: set to 1 if an instance names a mesh outside this call (or, in the
                               deterministic mode, if a workgroup stopped waiting for another one: see below) */
} vkr_cull_args;

/* Deterministic mode and other work on the GPU.  Its workgroups wait for status words of lower-numbered
 * workgroups of the same launch.  On a GPU that other streams or contexts keep busy some of them get their places
 * late, and the others wait for as long as that takes - the call gets slower, never wrong.  Only when an awaited
 * workgroup has not run for 10 s of wall clock (a device fault, foreign work that never ends) does a workgroup stop
 * waiting, so that the GPU is not held for ever; the launch's results are then invalid, error_flag (if given) is
 * set, and the next call on this context that waits for the stream - vkr_queue_wait_idle, vkr_buffer_download,
 * vkr_fence_wait, vkr_fence_status when signalled - returns VKR_ERR_DEVICE once, naming the launch. */
VKR_API int32_t vkr_prep_indirect(vkr_context* ctx, const vkr_cull_args* args);

/* cmd_out[m] = cmd_template[m] with instanceCount = 0 and firstInstance = first_instance + mesh_base[m], for
 * every mesh: the state vkr_prep_indirect (atomic mode) starts from, and what a frame loop such as
 * /root/reference/src/main.cpp:198-231 would record after its indirect draws (vkCmdFillBuffer over the counts)
 * so that the next frame's cull finds its counters at zero.  Only mesh_base, mesh_count, first_instance,
 * cmd_template and cmd_out of `args` are read. */
VKR_API int32_t vkr_prep_indirect_reset(vkr_context* ctx, const vkr_cull_args* args);

/* No counterpart in the reference.  For callers that cannot allocate device memory or let a library do it
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the deterministic
 * cull (visible[] ascending inside each mesh segment, as with VKR_CULL_DETERMINISTIC) with its state in a
 * fixed array that is part of the library's GPU code (32 KiB per GPU, held until the process ends; its
 * address is resolved once per GPU inside vkr_context_create).  Same arguments and results as
 * vkr_prep_indirect; VKR_CULL_DETERMINISTIC is implied.  The call makes no hipMalloc / hipFree or other
 * allocating call.  VKR_ERR_INVALID_ARGUMENT if instance_count > VKR_NOALLOC_MAX_INSTANCES.
 * All calls on one GPU share the one array, whatever their context or host thread: the library orders them (an
 * event recorded behind each launch and waited for, on the GPU, in front of the next), so they never overlap
 * each other; replays of a recorded call (below) are outside that ordering, and a replay that may run at the
 * same time as another user of the array on the same GPU races with it - order those yourself.
 * The array keeps the previous call's words; each call tags its words with a sequence number of its
 * own, so nothing has to be cleared (the library zeroes the array when the numbers wrap, after 2^30 calls).
 * Called while the context's stream (vkr_context_stream) is being captured into a HIP graph, the call records
 * two nodes: a clear of the array and the launch, tagged with a number reserved for recorded launches - a
 * replayed launch runs with the arguments it was recorded with and could not tell its own earlier words from
 * new ones otherwise.  Replays may be mixed freely with ordinary calls. */
#define VKR_NOALLOC_MAX_INSTANCES 4194304ull /* 2^22 */
VKR_API int32_t vkr_prep_indirect_noalloc(vkr_context* ctx, const vkr_cull_args* args);

/* ------------------------------------------------------------------ dispatch surface
 * Replaces the objects a caller of the reference touches to run a compute pass:
 * vw::Shader/loadShader (/root/reference/inc/vkshader.hpp:9-37) -> kernel looked up by name;
 * vw::ComputePipeline (/root/reference/inc/vkcompute.hpp:6-9) + vw::PipelineLayout
 * (/root/reference/inc/vkdescriptor.hpp:58-67) -> the kernel's fixed (set,binding) slot table;
 * vk::CommandBuffer::{bindPipeline,pushConstants,bindDescriptorSets,dispatch}
 * (/root/reference/src/main.cpp:222-226) -> vkr_cmd_*;
 * vw::Queue::oneTimeRecordSubmit + vw::Fence (/root/reference/inc/vkcore.hpp:34-48,140-152)
 * -> vkr_queue_submit + vkr_fence_*.
 *
 * Kernel "deferred": set 0 binding 0 elements 0..3 = albedo, material, normal, depth images;
 * set 0 binding 1 = light buffer (range / 48 = light count); set 1 binding 0 = output image;
 * push constants [0,80) = DeferredPushData, optional [80,96) = {flags, row_offset, full_height, 0};
 * dispatch(x, y, 1) with x, y = output extent (the reference dispatches one 1x1x1 group per pixel).
 * Kernel "prep_indirect": set 0 bindings 0..5 = aabb_min, aabb_max, mesh_base, cmd_template,
 * cmd_out, visible buffers; push constants [0,96) = planes,
 * [96,112) = {mesh_count, first_mesh, first_instance, flags}; dispatch(instance_count, 1, 1). */
VKR_API int32_t vkr_kernel_get(vkr_context* ctx, const char* name, vkr_kernel** out_kernel);
VKR_API int32_t vkr_kernel_push_constant_size(vkr_kernel* kernel, uint32_t* out_min_bytes, uint32_t* out_max_bytes);

VKR_API int32_t vkr_cmd_begin(vkr_context* ctx, vkr_cmd** out_cmd);
VKR_API int32_t vkr_cmd_bind_kernel(vkr_cmd* cmd, vkr_kernel* kernel);
/* Bytes are copied at record time, like vkCmdPushConstants.  offset+size <= 128. */
VKR_API int32_t vkr_cmd_push_constants(vkr_cmd* cmd, uint32_t offset, uint32_t size, const void* data);
VKR_API int32_t vkr_cmd_bind_buffer(vkr_cmd* cmd, uint32_t set, uint32_t binding, vkr_buffer* buf, uint64_t offset,
                                    uint64_t range);
VKR_API int32_t vkr_cmd_bind_image2d(vkr_cmd* cmd, uint32_t set, uint32_t binding, uint32_t array_element,
                                     vkr_buffer* buf, uint64_t offset, uint32_t width, uint32_t height,
                                     uint32_t pitch_bytes, uint32_t format);
VKR_API int32_t vkr_cmd_dispatch(vkr_cmd* cmd, uint32_t x, uint32_t y, uint32_t z);
VKR_API int32_t vkr_cmd_end(vkr_cmd* cmd);
VKR_API int32_t vkr_cmd_destroy(vkr_cmd* cmd);

/* Replays an ended command list on the context's stream and returns a fence that signals when it
 * has finished.  A list may be submitted repeatedly. */
VKR_API int32_t vkr_queue_submit(vkr_context* ctx, vkr_cmd* cmd, vkr_fence** out_fence);
/* A fence for "everything submitted so far" (used for timing brackets). */
VKR_API int32_t vkr_queue_marker(vkr_context* ctx, vkr_fence** out_fence);
VKR_API int32_t vkr_queue_wait_idle(vkr_context* ctx);
/* VKR_OK once signalled, VKR_ERR_NOT_READY otherwise (vw::Fence::signaled). */
VKR_API int32_t vkr_fence_status(vkr_fence* fence);
/* Blocks up to timeout_ns (UINT64_MAX = forever); VKR_ERR_NOT_READY on timeout (vw::Fence::wait). */
VKR_API int32_t vkr_fence_wait(vkr_fence* fence, uint64_t timeout_ns);
/* Milliseconds of GPU time between two signalled fences of one context. */
VKR_API int32_t vkr_fence_elapsed_ms(vkr_fence* begin, vkr_fence* end, float* out_ms);
VKR_API int32_t vkr_fence_destroy(vkr_fence* fence);

/* ------------------------------------------------------------------ host math
 * The matrices the reference's caller builds with glm before each dispatch
 * (/root/reference/src/main.cpp:95,206-209; /root/reference/inc/vkcamera.hpp:51-53).  All 4x4
 * matrices are column-major (m[col*4+row]), binary32. */
VKR_API void vkr_perspective(float fovy_radians, float aspect, float z_near, float z_far, float out_m[16]);
VKR_API void vkr_look_at(const float eye[3], const float center[3], const float up[3], float out_m[16]);
VKR_API void vkr_mat4_mul(const float a[16], const float b[16], float out_m[16]);
/* VKR_ERR_INVALID_ARGUMENT (out_m untouched) if the matrix is singular. */
VKR_API int32_t vkr_mat4_inverse(const float m[16], float out_m[16]);
/* Six un-normalised clip planes of a view-projection matrix under Vulkan clip rules
 * (-w <= x,y <= w, 0 <= z <= w; /root/reference/shaders/offscreen.vert:31):
 * row3+row0, row3-row0, row3+row1, row3-row1, row2, row3-row2. */
VKR_API void vkr_frustum_planes(const float view_proj[16], float out_planes[24]);

/* ------------------------------------------------------------------ camera and per-frame constants
 * CameraInputHandler of /root/reference/inc/vkcamera.hpp:9-67 without the GLFW types: the state
 * (:62-63), getView (:51-53), the key movement of onKeyEvent (:30-47) and the mouse look of
 * onCursorPosChange (:11-29). */
typedef struct vkr_camera {
  float pos[3];     /* default (0, 3, 71) */
  float forward[3]; /* default (0, 0, -1) */
  float up[3];      /* constant (0, -1, 0) */
} vkr_camera;

enum { VKR_KEY_W = 0, VKR_KEY_A = 1, VKR_KEY_S = 2, VKR_KEY_D = 3, VKR_KEY_SPACE = 4, VKR_KEY_LEFT_SHIFT = 5 };

VKR_API void vkr_camera_default(vkr_camera* cam);
/* lookAt(pos, pos + forward, up). */
VKR_API void vkr_camera_view(const vkr_camera* cam, float out_view[16]);
/* pos += dir * (0.2 * delta_time), negated for SPACE / A / S; dir = cross(forward, up) for A / D, up for
 * SPACE / LEFT_SHIFT, forward for W / S.  delta_time is the reference's mDeltaTime (frame time in
 * units of 10 ms).  Unknown keys are ignored. */
VKR_API void vkr_camera_key(vkr_camera* cam, int32_t key, float delta_time);
/* Mouse look by a cursor delta (pixels): 60 degrees per unit of delta, pitch clamped to
 * [0.01 pi, 0.99 pi] from the up vector. */
VKR_API void vkr_camera_cursor_delta(vkr_camera* cam, float dx, float dy);

/* What the frame callback of /root/reference/src/main.cpp:206-209 computes before recording: view,
 * vp = proj * view, inverse(vp), DeferredPushData {cameraPos, 1.0f, inverse(vp)}; plus the clip
 * planes the cull pass takes. */
typedef struct vkr_frame_constants {
  float view[16];
  float view_proj[16];
  float inv_view_proj[16];
  float planes[24];
  float deferred_push[20];
} vkr_frame_constants;
/* VKR_ERR_INVALID_ARGUMENT if vp is singular. */
VKR_API int32_t vkr_frame_setup(const vkr_camera* cam, const float proj[16], vkr_frame_constants* out);

/* ------------------------------------------------------------------ light table source
 * LightInfo of /root/reference/src/main.cpp:40-47 (= the std140 Light of deferred.comp:13-19). */
typedef struct vkr_light_info {
  float pos[3];
  float opening_angle;  /* cos(radians(opening_angle) / 2) */
  float color[3];       /* "intensity" */
  float penumbra_angle; /* cos(radians(penumbra_angle) / 2) */
  float dir[3];
  float _temp;
} vkr_light_info;

/* The per-light arithmetic of /root/reference/src/main.cpp:57-69 over arrays of the JSON fields
 * (angles in degrees). */
VKR_API void vkr_pack_point_lights(uint32_t count, const float* pos3, const float* intensity3, const float* direction3,
                                   const float* opening_angle_deg, const float* penumbra_angle_deg, vkr_light_info* out);
/* Parses the text of a .fscene file (JSON) and packs every entry of "lights" whose "type" is
 * "point_light" (/root/reference/src/main.cpp:48-70).  out may be null to count only.
 * VKR_ERR_INVALID_ARGUMENT on malformed JSON or a point light missing a field the reference reads;
 * VKR_ERR_OUT_OF_MEMORY if capacity is too small (out_count still holds the number found). */
VKR_API int32_t vkr_fscene_point_lights(const char* json_text, uint64_t json_bytes, vkr_light_info* out, uint32_t capacity,
                                        uint32_t* out_count);

/* ------------------------------------------------------------------ scene tables
 * What vw::Scene::Scene builds from an imported scene before any pass runs
 * (/root/reference/src/vkmodel.cpp:55-134): the node tree flattened into per-mesh instance matrix
 * lists (:55-73), then per kept mesh a MeshInfo, a template draw command, a PerMeshData and its
 * contiguous run of model matrices (:89-111).  The description replaces the aiScene fields those
 * lines read. */
#define VKR_MESH_HAS_FACES 1u
#define VKR_MESH_HAS_UV0 2u
#define VKR_MESH_HAS_TRIANGLES 4u /* a mesh lacking any of the three is skipped (:91-93) */

typedef struct vkr_scene_desc {
  uint32_t node_count;              /* node 0 is the root */
  const float* node_transform;      /* node_count x 16: aiMatrix4x4 a1..d4 (row-major), :61-65 */
  const uint32_t* node_child_begin; /* node_count + 1 offsets into node_children */
  const uint32_t* node_children;    /* child node indices, in aiNode::mChildren order */
  const uint32_t* node_mesh_begin;  /* node_count + 1 offsets into node_meshes */
  const uint32_t* node_meshes;      /* mesh indices (aiNode::mMeshes) */
  uint32_t mesh_count;
  const uint32_t* mesh_vertex_count; /* aiMesh::mNumVertices */
  const uint32_t* mesh_face_count;   /* aiMesh::mNumFaces */
  const uint32_t* mesh_material;     /* aiMesh::mMaterialIndex */
  const uint32_t* mesh_flags;        /* VKR_MESH_* */
} vkr_scene_desc;

typedef struct vkr_scene_counts {
  uint32_t kept_meshes;     /* meshes that pass :91-93 */
  uint32_t kept_instances;  /* model matrices emitted (instances of kept meshes) */
  uint32_t total_instances; /* mTotalInstanceCount: every node-mesh reference, kept or not (:68) */
  uint32_t total_vertices;  /* mTotalVertexCount */
  uint32_t total_indices;   /* mTotalIndexCount */
} vkr_scene_counts;

typedef struct vkr_mesh_info { /* vw::MeshInfo, /root/reference/inc/vkmodel.hpp:29-34 */
  uint32_t indexCount, firstIndex, vertexOffset, materialIndex;
} vkr_mesh_info;

typedef struct vkr_per_mesh_data { /* /root/reference/inc/vkmodel.hpp:76-78 */
  uint32_t materialIdx, modelMatrixBaseIndex;
} vkr_per_mesh_data;

typedef struct vkr_scene_tables { /* caller-allocated, sized from vkr_scene_counts */
  vkr_mesh_info* mesh_info;                         /* kept_meshes */
  vkr_draw_indexed_indirect_command* draw_commands; /* kept_meshes: {3*faces, #instances, firstIndex, vertexOffset, 0} */
  vkr_per_mesh_data* per_mesh_data;                 /* kept_meshes */
  float* mesh_matrices;                             /* kept_instances x 16, column-major */
  uint32_t* mesh_base;                              /* kept_meshes + 1: modelMatrixBaseIndex per mesh, then the total */
  uint32_t* vertex_base;                            /* kept_meshes + 1: vertexOffset per mesh, then the total */
  uint32_t* source_mesh;                            /* kept_meshes: index of the mesh in the description */
} vkr_scene_tables;

VKR_API int32_t vkr_scene_count(const vkr_scene_desc* desc, vkr_scene_counts* out_counts);
VKR_API int32_t vkr_scene_build_tables(const vkr_scene_desc* desc, const vkr_scene_counts* counts, vkr_scene_tables* out);

/* Object-space bounds of every kept mesh: the min / max of computeAABB
 * (/root/reference/src/vkmodel.cpp:166-171) over the mesh's vertices.  positions = the packed
 * float3 position stream of the vertex buffer (:116), vertex_base as in vkr_scene_tables; mesh_aabb
 * receives per mesh two float4 (min.xyz, 0) (max.xyz, 0).  A mesh without vertices gets
 * (+inf, -inf).  All pointers are device memory. */
VKR_API int32_t vkr_mesh_aabbs(vkr_context* ctx, const void* positions, const uint32_t* vertex_base, uint32_t mesh_count,
                               void* mesh_aabb);
/* World-space AABB of every instance, in the layout vkr_prep_indirect reads: the eight corners of the
 * mesh's bounds in the order of /root/reference/src/vkmodel.cpp:172-173, each mapped by the
 * instance's model matrix as ((m0*x + m1*y) + m2*z) + m3 per row in binary32 without fused
 * multiply-add, then min / max.  aabb_min[i].w = mesh id of instance i.  All pointers device memory;
 * mesh_matrices 16-byte aligned. */
VKR_API int32_t vkr_instance_aabbs(vkr_context* ctx, const void* mesh_matrices, const uint32_t* mesh_base, const void* mesh_aabb,
                                   uint32_t mesh_count, uint64_t instance_count, void* aabb_min, void* aabb_max);

/* ------------------------------------------------------------------ G-buffer producers
 * The four attachments the raster pass of /root/reference/src/main.cpp:111-129,214-219 fills and
 * the deferred pass reads. */

/* Attachment write conversions of /root/reference/shaders/offscreen.frag:13-16 into the formats of
 * /root/reference/src/main.cpp:112-116: float4 -> R8G8B8A8_UNORM (NaN -> 0, clamp, round half to
 * even), float4 -> R16G16B16A16_SFLOAT (round to nearest even), float -> D32_SFLOAT.  The four
 * sources are device arrays of width x height float4 / float, rows contiguous. */
typedef struct vkr_gbuffer_encode_args {
  const void* albedo_f32;   /* float4 per pixel */
  const void* material_f32; /* float4 per pixel */
  const void* normal_f32;   /* float4 per pixel */
  const void* depth_f32;    /* float per pixel */
  vkr_image2d albedo, material, normal, depth; /* destinations */
} vkr_gbuffer_encode_args;
VKR_API int32_t vkr_gbuffer_encode(vkr_context* ctx, const vkr_gbuffer_encode_args* args);

/* Both stages of the reference's G-buffer pass for one surface sample per pixel (a software producer that keeps
 * the raster pass's interpolated attributes per pixel, visibility-buffer style):
 *   /root/reference/shaders/offscreen.vert:28-35  M = modelMatrices[index]; N = normalize(vec3(M * vec4(inNormal, 0)));
 *                                                 T = normalize(vec3(M * vec4(inTangent, 0))); B = cross(N, T);
 *   /root/reference/shaders/offscreen.frag:13-16  outColor, outSpecular = the two texels; outNormal =
 *                                                 vec4(mat3(T, B, N) * normalize(normalTexel.xyz * 2 - 1), 1),
 * then the attachment write conversions of vkr_gbuffer_encode.  Arithmetic, binary32 without fused multiply-add,
 * divisions and square roots correctly rounded: vec3(M * vec4(v, 0)) = (M[0]*v.x + M[1]*v.y) + M[2]*v.z per row
 * (the w = 0 term is dropped); normalize(v) = v / sqrt((v.x*v.x + v.y*v.y) + v.z*v.z); cross as written in GLSL;
 * mat3 * v = (T*v.x + B*v.y) + N*v.z.  Where the raster pass normalises N and T per vertex and interpolates
 * the matrix, this call builds the matrix per pixel from the pixel's own attributes.
 * Sources: device arrays of width x height float4 / float, rows contiguous, 16-byte aligned.
 * normal_obj[p].w holds the pixel's index into model_matrices as uint32 bits (an index >= matrix_count reads
 * matrix 0); model_matrices = column-major mat4 array, 16-byte aligned (the array of
 * /root/reference/src/vkmodel.cpp:108-110 that vkr_scene_build_tables fills). */
typedef struct vkr_gbuffer_surface_args {
  const void* albedo_texel;   /* float4 per pixel: texture(samplers[texOffset], uv) */
  const void* specular_texel; /* float4 per pixel: texture(samplers[texOffset + 2], uv) */
  const void* normal_texel;   /* float4 per pixel: texture(samplers[texOffset + 1], uv), xyz in [0, 1] */
  const void* normal_obj;     /* float4 per pixel: inNormal.xyz, w = model matrix index */
  const void* tangent_obj;    /* float4 per pixel: inTangent.xyz, w ignored */
  const void* depth_f32;      /* float per pixel */
  const void* model_matrices; /* matrix_count x 16 floats */
  uint32_t matrix_count;
  uint32_t reserved;
  vkr_image2d albedo, material, normal, depth; /* destinations */
} vkr_gbuffer_surface_args;
VKR_API int32_t vkr_gbuffer_encode_surfaces(vkr_context* ctx, const vkr_gbuffer_surface_args* args);

/* An analytic test scene (spheres over a checkered ground plane) written straight into the four
 * attachments: per pixel the nearest hit along the camera ray through the pixel centre, depth =
 * clip.z / clip.w of the hit under view_proj (1.0 where nothing is hit, the clear value of
 * /root/reference/src/main.cpp:129), normal = minus the outward surface normal (the deferred pass
 * negates it, /root/reference/shaders/deferred.comp:108), material = (occlusion, roughness,
 * metalness, 1).  Not a rasteriser: it exists to give the deferred pass physically coherent inputs. */
typedef struct vkr_sphere {
  float center[3];
  float radius;
  float albedo[3];
  float roughness;
  float metalness;
  float occlusion;
  float pad[2];
} vkr_sphere;

typedef struct vkr_gbuffer_scene {
  float view_proj[16];
  float inv_view_proj[16];
  float camera_pos[3];
  uint32_t sphere_count;
  float ground_plane[4];    /* n.xyz, d: points with dot(n, p) + d = 0; n = 0 disables the ground */
  float ground_albedo_a[3]; /* checker colours */
  float ground_tile;        /* checker edge length (world units), > 0 */
  float ground_albedo_b[3];
  float ground_roughness;
  float ground_metalness;
  float pad[3];
} vkr_gbuffer_scene;

typedef struct vkr_gbuffer_fill_args {
  vkr_image2d albedo, material, normal, depth; /* destinations */
  const void* spheres;                         /* device: scene.sphere_count x vkr_sphere */
  vkr_gbuffer_scene scene;
  uint32_t row_offset;  /* the images hold rows [row_offset, row_offset + height) of a frame */
  uint32_t full_height; /* full_height rows tall; 0 / 0 = whole frame */
} vkr_gbuffer_fill_args;
VKR_API int32_t vkr_gbuffer_fill(vkr_context* ctx, const vkr_gbuffer_fill_args* args);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* VKR_C_H_ */
