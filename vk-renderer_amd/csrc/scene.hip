// Scene tables on the device: object-space bounds per mesh and world-space bounds per instance, the
// inputs of the cull pass (the step before it in the frame).
//   mesh bounds     = min / max of computeAABB, /root/reference/src/vkmodel.cpp:166-171
//   instance bounds = the eight corners of :172-173 under the instance's model matrix
//                     (/root/reference/shaders/offscreen.vert:28-31 applies the same matrix), min / max.
// Both are HBM-bound streams.  Contraction is off for this file: a corner coordinate is
// ((m0*x + m1*y) + m2*z) + m3 in binary32, the same bits as the CPU restatement.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vkr_c.h"
#include "vkr_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_stream(const float4* p) {
  const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void store_stream(float4* p, float4 v) {
  v4f t;
  t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

constexpr int kMeshBlock = 256;

// One workgroup per mesh: strided pass over the mesh's packed float3 positions, wave reduction,
// four partials through LDS.  min / max are exact, so the result does not depend on the order.
__global__ __launch_bounds__(kMeshBlock) void mesh_aabb_kernel(const float* __restrict__ positions, const uint32_t* __restrict__ vertex_base,
                                                               float4* __restrict__ mesh_aabb) {
  const uint32_t m = blockIdx.x;
  const uint32_t v0 = vertex_base[m], v1 = vertex_base[m + 1];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t v = v0 + threadIdx.x; v < v1; v += kMeshBlock) {
    const float* p = positions + 3ull * v;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x = p[c];
      mn[c] = x < mn[c] ? x : mn[c];  // NaN never replaces a bound, as glm::min(pos, min) keeps `min`
      mx[c] = x > mx[c] ? x : mx[c];
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float a = __shfl_xor(mn[c], off), b = __shfl_xor(mx[c], off);
      mn[c] = a < mn[c] ? a : mn[c];
      mx[c] = b > mx[c] ? b : mx[c];
    }
  }
  __shared__ float s_part[kMeshBlock / 64][6];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  if (lane == 0u) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_part[wave][c] = mn[c];
      s_part[wave][3 + c] = mx[c];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0u) {
#pragma unroll
    for (int w = 1; w < kMeshBlock / 64; ++w) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        mn[c] = s_part[w][c] < mn[c] ? s_part[w][c] : mn[c];
        mx[c] = s_part[w][3 + c] > mx[c] ? s_part[w][3 + c] : mx[c];
      }
    }
    mesh_aabb[2u * m] = make_float4(mn[0], mn[1], mn[2], 0.0f);
    mesh_aabb[2u * m + 1u] = make_float4(mx[0], mx[1], mx[2], 0.0f);
  }
}

// Largest m with mesh_base[m] <= i (instances of a mesh are contiguous; empty meshes are skipped).
__device__ __forceinline__ uint32_t mesh_of(const uint32_t* __restrict__ mesh_base, uint32_t mesh_count, uint32_t i) {
  uint32_t lo = 0, hi = mesh_count;  // mesh_base[lo] <= i < mesh_base[hi]
  while (hi - lo > 1u) {
    const uint32_t mid = (lo + hi) >> 1;
    if (mesh_base[mid] <= i)
      lo = mid;
    else
      hi = mid;
  }
  return lo;
}

constexpr int kInstBlock = 256;

// One lane per instance.  The 64-byte matrices of a wave are one contiguous 4 KiB run: they are
// fetched with fully coalesced 16-byte loads (lane l takes float4 number l, l+256, ... of the block's run)
// into LDS and each lane then reads its own four columns back.  The mesh lookup runs while those loads
// are in flight: one binary search per workgroup for its first instance (uniform, so it runs on scalar
// loads), then each lane steps forward over the few mesh boundaries inside the workgroup's 256 instances.
__global__ __launch_bounds__(kInstBlock) void instance_aabb_kernel(const float4* __restrict__ matrices, const uint32_t* __restrict__ mesh_base,
                                                                   const float4* __restrict__ mesh_aabb, uint32_t mesh_count, uint32_t instance_count,
                                                                   float4* __restrict__ aabb_min, float4* __restrict__ aabb_max) {
  // Column-major staging: column c of instance i at s_mat[c * kColStride + i].  The reads (lane = instance) are then
  // consecutive float4; the writes (lane k holds column k & 3 of instance k >> 2) put the 16 lanes that share an
  // LDS pass at word 4*i + 16*c of the 64 banks, i, c = 0..3: distinct, thanks to the 4 float4 of padding per
  // column.  SQ_LDS_BANK_CONFLICT still reads a third of SQ_LDS_IDX_ACTIVE for this kernel, exactly as it did for a
  // row-major staging with other padding: the counter charges every 128-bit access of a 64-lane wave its extra
  // passes (1 KiB through a 256-byte-per-clock LDS), whatever the banks.
  constexpr uint32_t kColStride = kInstBlock + 4;
  __shared__ float4 s_mat[4 * kColStride];
  const uint32_t block_first = blockIdx.x * kInstBlock;
  const uint32_t in_block = min((uint32_t)kInstBlock, instance_count - block_first);
  const float4* src = matrices + 4ull * block_first;
  const uint32_t t = threadIdx.x;
  float4 staged[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t k = t + (uint32_t)j * kInstBlock;
    if (k < 4u * in_block) staged[j] = load_stream(src + k);
  }
  const uint32_t i = block_first + t;
  uint32_t m = mesh_of(mesh_base, mesh_count, block_first);
  if (t < in_block) {
    while (m + 1u < mesh_count && mesh_base[m + 1u] <= i) ++m;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t k = t + (uint32_t)j * kInstBlock;
    if (k < 4u * in_block) s_mat[(k & 3u) * kColStride + (k >> 2)] = staged[j];
  }
  __syncthreads();
  if (t >= in_block) return;
  const float4 lo = mesh_aabb[2u * m], hi = mesh_aabb[2u * m + 1u];
  const float4 c0 = s_mat[t], c1 = s_mat[kColStride + t], c2 = s_mat[2u * kColStride + t], c3 = s_mat[3u * kColStride + t];
  const float m0[3] = {c0.x, c0.y, c0.z}, m1[3] = {c1.x, c1.y, c1.z}, m2[3] = {c2.x, c2.y, c2.z}, m3[3] = {c3.x, c3.y, c3.z};
  // corner order of /root/reference/src/vkmodel.cpp:172-173
  const float cx[8] = {lo.x, hi.x, lo.x, lo.x, hi.x, hi.x, lo.x, hi.x};
  const float cy[8] = {lo.y, lo.y, hi.y, lo.y, hi.y, lo.y, hi.y, hi.y};
  const float cz[8] = {lo.z, lo.z, lo.z, hi.z, lo.z, hi.z, hi.z, hi.z};
  float mn[3], mx[3];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float w = ((m0[r] * cx[k] + m1[r] * cy[k]) + m2[r] * cz[k]) + m3[r];
      if (k == 0) {
        mn[r] = w;
        mx[r] = w;
      } else {
        mn[r] = w < mn[r] ? w : mn[r];
        mx[r] = w > mx[r] ? w : mx[r];
      }
    }
  }
  store_stream(aabb_min + i, make_float4(mn[0], mn[1], mn[2], __uint_as_float(m)));
  store_stream(aabb_max + i, make_float4(mx[0], mx[1], mx[2], 0.0f));
}

}  // namespace

extern "C" {

int32_t vkr_mesh_aabbs(vkr_context* ctx, const void* positions, const uint32_t* vertex_base, uint32_t mesh_count, void* mesh_aabb) {
  if (!ctx) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_mesh_aabbs: null context");
  if (mesh_count == 0u) return VKR_OK;
  if (!vertex_base || !mesh_aabb) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_mesh_aabbs: null vertex_base / mesh_aabb");
  if (reinterpret_cast<uintptr_t>(mesh_aabb) & 15u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_mesh_aabbs: mesh_aabb must be 16-byte aligned");
  if (reinterpret_cast<uintptr_t>(positions) & 3u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_mesh_aabbs: positions must be 4-byte aligned");
  VKR_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(mesh_aabb_kernel, dim3(mesh_count), dim3(kMeshBlock), 0, ctx->stream, static_cast<const float*>(positions), vertex_base,
                     static_cast<float4*>(mesh_aabb));
  VKR_HIP(hipGetLastError());
  return VKR_OK;
}

int32_t vkr_instance_aabbs(vkr_context* ctx, const void* mesh_matrices, const uint32_t* mesh_base, const void* mesh_aabb, uint32_t mesh_count,
                           uint64_t instance_count, void* aabb_min, void* aabb_max) {
  if (!ctx) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_instance_aabbs: null context");
  if (instance_count == 0u) return VKR_OK;
  if (instance_count >= (1ull << 31)) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "instance_count %llu >= 2^31", (unsigned long long)instance_count);
  if (mesh_count == 0u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_instance_aabbs: instances without meshes");
  if (!mesh_matrices || !mesh_base || !mesh_aabb || !aabb_min || !aabb_max) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_instance_aabbs: null pointer");
  if ((reinterpret_cast<uintptr_t>(mesh_matrices) | reinterpret_cast<uintptr_t>(mesh_aabb) | reinterpret_cast<uintptr_t>(aabb_min) |
       reinterpret_cast<uintptr_t>(aabb_max)) & 15u)
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_instance_aabbs: mesh_matrices / mesh_aabb / aabb_min / aabb_max must be 16-byte aligned");
  VKR_HIP(hipSetDevice(ctx->device));
  const uint32_t n = (uint32_t)instance_count;
  hipLaunchKernelGGL(instance_aabb_kernel, dim3((n + kInstBlock - 1) / kInstBlock), dim3(kInstBlock), 0, ctx->stream,
                     static_cast<const float4*>(mesh_matrices), mesh_base, static_cast<const float4*>(mesh_aabb), mesh_count, n,
                     static_cast<float4*>(aabb_min), static_cast<float4*>(aabb_max));
  VKR_HIP(hipGetLastError());
  return VKR_OK;
}

}  // extern "C"
