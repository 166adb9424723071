// Producers of the four G-buffer attachments the deferred pass reads
// (/root/reference/src/main.cpp:111-117: albedo rgba8, material rgba8, normal rgba16f, depth d32f).
//   encode: the attachment write conversions of /root/reference/shaders/offscreen.frag:13-16 applied
//           to float planes (what the ROPs do in the reference's raster pass);
//   surfaces: both shader stages of that pass per pixel (offscreen.vert:28-35 builds the tangent frame from the
//           pixel's model matrix, offscreen.frag:13-16 applies the normal map), then the same conversions;
//   fill:   an analytic scene (spheres over a checkered plane) ray-cast per pixel and written in those
//           formats - test and bench input with coherent depth / normals, not a rasteriser.
// Both are write-bound streams (20 bytes per pixel).  Contraction is off for this file and every
// division / square root is the correctly rounded one, so the bytes written equal those of a binary32
// CPU evaluation of the same expressions (sqrtf, not __fsqrt_rn: the latter is the approximate native square root).
#include <hip/hip_fp16.h>
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

__device__ __forceinline__ uint32_t unorm8(float c) {
  // NaN -> 0, clamp to [0,1], scale, round half to even (Vulkan float -> UNORM conversion)
  const float t = (c > 0.0f) ? (c < 1.0f ? c : 1.0f) : 0.0f;
  return (uint32_t)__builtin_rintf(t * 255.0f);
}
__device__ __forceinline__ uint32_t pack_unorm8x4(float r, float g, float b, float a) {
  return unorm8(r) | (unorm8(g) << 8) | (unorm8(b) << 16) | (unorm8(a) << 24);
}
__device__ __forceinline__ uint2 pack_half4(float x, float y, float z, float w) {
  const __half2 lo = __floats2half2_rn(x, y), hi = __floats2half2_rn(z, w);
  return make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
}

struct Planes {
  uint8_t *albedo, *material, *normal, *depth;
  uint32_t pitch_albedo, pitch_material, pitch_normal, pitch_depth;
  uint32_t width, height;
};

// ---------------------------------------------------------------- encode
// A 256-thread workgroup converts 1024 consecutive pixels of one row: lane t takes pixels t, t+256,
// t+512, t+768 of that run, so every load is one fully used 16-byte lane slot (4 KiB per wave and
// instruction) and every store a contiguous dword / qword per lane.
__global__ __launch_bounds__(256) void gbuffer_encode_kernel(const float4* __restrict__ albedo, const float4* __restrict__ material,
                                                             const float4* __restrict__ normal, const float* __restrict__ depth, Planes o) {
  const uint32_t y = blockIdx.y;
  const uint32_t x_first = blockIdx.x * 1024u + threadIdx.x;
  const size_t row = (size_t)y * o.width;
  float4 av[4], mv[4], nv[4];
  float dv[4];
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t x = x_first + i * 256u;
    const size_t p = row + (x < o.width ? x : 0u);
    av[i] = load_stream(albedo + p);
    mv[i] = load_stream(material + p);
    nv[i] = load_stream(normal + p);
    dv[i] = __builtin_nontemporal_load(depth + p);
  }
  uint32_t* pa = reinterpret_cast<uint32_t*>(o.albedo + (size_t)y * o.pitch_albedo);
  uint32_t* pm = reinterpret_cast<uint32_t*>(o.material + (size_t)y * o.pitch_material);
  uint2* pn = reinterpret_cast<uint2*>(o.normal + (size_t)y * o.pitch_normal);
  float* pd = reinterpret_cast<float*>(o.depth + (size_t)y * o.pitch_depth);
#pragma unroll
  for (uint32_t i = 0; i < 4u; ++i) {
    const uint32_t x = x_first + i * 256u;
    if (x < o.width) {
      pa[x] = pack_unorm8x4(av[i].x, av[i].y, av[i].z, av[i].w);
      pm[x] = pack_unorm8x4(mv[i].x, mv[i].y, mv[i].z, mv[i].w);
      pn[x] = pack_half4(nv[i].x, nv[i].y, nv[i].z, nv[i].w);
      pd[x] = dv[i];
    }
  }
}

// ---------------------------------------------------------------- surfaces: offscreen.vert:28-35 + offscreen.frag:13-16 per pixel
struct Vec3 {
  float x, y, z;
};
// vec3(M * vec4(v, 0)) with M's first three columns c0, c1, c2
__device__ __forceinline__ Vec3 rotate_dir(const float4 c0, const float4 c1, const float4 c2, const float4 v) {
  return {(c0.x * v.x + c1.x * v.y) + c2.x * v.z, (c0.y * v.x + c1.y * v.y) + c2.y * v.z, (c0.z * v.x + c1.z * v.y) + c2.z * v.z};
}
__device__ __forceinline__ Vec3 normalize3(const Vec3 v) {
  const float len = sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z);
  return {v.x / len, v.y / len, v.z / len};
}

struct SurfaceSources {
  const float4 *albedo, *specular, *normal_texel, *normal_obj, *tangent_obj;
  const float* depth;
  const float4* matrices;
  uint32_t matrix_count;
};

// A 256-thread workgroup takes 512 consecutive pixels of one row, lane t the pixels t and t + 256: every load is a
// fully used 16-byte lane slot.  84 bytes read and 20 written per pixel, plus the pixel's matrix (48 bytes of it;
// neighbouring pixels name the same one, so those come from the caches): a stream, bound by HBM.
__global__ __launch_bounds__(256) void gbuffer_surface_kernel(const SurfaceSources s, Planes o) {
  const uint32_t y = blockIdx.y;
  const uint32_t x_first = blockIdx.x * 512u + threadIdx.x;
  const size_t row = (size_t)y * o.width;
  float4 av[2], sv[2], tv[2], nv[2], gv[2];
  float dv[2];
#pragma unroll
  for (uint32_t i = 0; i < 2u; ++i) {
    const uint32_t x = x_first + i * 256u;
    const size_t p = row + (x < o.width ? x : 0u);
    nv[i] = load_stream(s.normal_obj + p);
    gv[i] = load_stream(s.tangent_obj + p);
    tv[i] = load_stream(s.normal_texel + p);
    av[i] = load_stream(s.albedo + p);
    sv[i] = load_stream(s.specular + p);
    dv[i] = __builtin_nontemporal_load(s.depth + p);
  }
  uint32_t* pa = reinterpret_cast<uint32_t*>(o.albedo + (size_t)y * o.pitch_albedo);
  uint32_t* pm = reinterpret_cast<uint32_t*>(o.material + (size_t)y * o.pitch_material);
  uint2* pn = reinterpret_cast<uint2*>(o.normal + (size_t)y * o.pitch_normal);
  float* pd = reinterpret_cast<float*>(o.depth + (size_t)y * o.pitch_depth);
#pragma unroll
  for (uint32_t i = 0; i < 2u; ++i) {
    const uint32_t x = x_first + i * 256u;
    uint32_t mi = __float_as_uint(nv[i].w);
    if (mi >= s.matrix_count) mi = 0u;
    const float4 c0 = s.matrices[4u * (size_t)mi], c1 = s.matrices[4u * (size_t)mi + 1u], c2 = s.matrices[4u * (size_t)mi + 2u];
    const Vec3 N = normalize3(rotate_dir(c0, c1, c2, nv[i]));  // offscreen.vert:32
    const Vec3 T = normalize3(rotate_dir(c0, c1, c2, gv[i]));  // :33
    const Vec3 B = {N.y * T.z - N.z * T.y, N.z * T.x - N.x * T.z, N.x * T.y - N.y * T.x};  // :34
    const Vec3 m = normalize3({tv[i].x * 2.0f - 1.0f, tv[i].y * 2.0f - 1.0f, tv[i].z * 2.0f - 1.0f});  // offscreen.frag:15
    const float nx = (T.x * m.x + B.x * m.y) + N.x * m.z;
    const float ny = (T.y * m.x + B.y * m.y) + N.y * m.z;
    const float nz = (T.z * m.x + B.z * m.y) + N.z * m.z;
    if (x < o.width) {
      pa[x] = pack_unorm8x4(av[i].x, av[i].y, av[i].z, av[i].w);  // :13
      pm[x] = pack_unorm8x4(sv[i].x, sv[i].y, sv[i].z, sv[i].w);  // :14
      pn[x] = pack_half4(nx, ny, nz, 1.0f);                       // :16
      pd[x] = dv[i];
    }
  }
}

// ---------------------------------------------------------------- fill
struct FillParams {
  Planes o;
  const vkr_sphere* spheres;
  uint32_t row_offset, full_height;
  vkr_gbuffer_scene scene;
};

constexpr int kSphereChunk = 128;

struct Hit {
  float t;        // ray parameter in units of (far point - camera); 1 = far plane
  int32_t what;   // -1 none, -2 ground, >= 0 sphere index
};

__global__ __launch_bounds__(256) void gbuffer_fill_kernel(const FillParams p) {
  __shared__ float4 s_sphere[kSphereChunk];  // camera - centre, |camera - centre|^2 - radius^2
  __shared__ uint32_t s_index[kSphereChunk];  // table index of each staged sphere
  __shared__ uint32_t s_kept[2];
  const uint32_t t = threadIdx.x;
  const uint32_t x0 = blockIdx.x * 32u + (t & 7u) * 4u;
  const uint32_t y = blockIdx.y * 32u + (t >> 3);
  const bool inside = (x0 < p.o.width) && (y < p.o.height);
  const vkr_gbuffer_scene& sc = p.scene;
  const float* M = sc.inv_view_proj;
  const float cam[3] = {sc.camera_pos[0], sc.camera_pos[1], sc.camera_pos[2]};

  float dir[4][3], a[4];  // ray direction (far point - camera) and its squared length
  Hit hit[4];
  const float v = ((float)(y + p.row_offset) + 0.5f) / (float)p.full_height;
  const float cy = v * 2.0f - 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float u = ((float)(x0 + i) + 0.5f) / (float)p.o.width;
    const float cx = u * 2.0f - 1.0f;
    // far-plane point of the pixel: inverse(VP) * (cx, cy, 1, 1), then the perspective divide
    const float wx = ((M[0] * cx + M[4] * cy) + M[8]) + M[12];
    const float wy = ((M[1] * cx + M[5] * cy) + M[9]) + M[13];
    const float wz = ((M[2] * cx + M[6] * cy) + M[10]) + M[14];
    const float ww = ((M[3] * cx + M[7] * cy) + M[11]) + M[15];
    dir[i][0] = wx / ww - cam[0];
    dir[i][1] = wy / ww - cam[1];
    dir[i][2] = wz / ww - cam[2];
    hit[i].t = 1.0f;
    hit[i].what = -1;
    a[i] = (dir[i][0] * dir[i][0] + dir[i][1] * dir[i][1]) + dir[i][2] * dir[i][2];
  }

  // The four side faces of the tile's ray pyramid (apex = camera, through the far-plane points of the tile's
  // corners; the rays of the tile's pixel centres lie inside it): outward unit normals, the same in every thread.
  float face[4][3];
  {
    float corner[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t ex = blockIdx.x * 32u + ((k == 1 || k == 2) ? 32u : 0u);
      const uint32_t ey = blockIdx.y * 32u + p.row_offset + (k >= 2 ? 32u : 0u);
      const float cx = (float)ex / (float)p.o.width * 2.0f - 1.0f;
      const float cyk = (float)ey / (float)p.full_height * 2.0f - 1.0f;
      const float wx = ((M[0] * cx + M[4] * cyk) + M[8]) + M[12];
      const float wy = ((M[1] * cx + M[5] * cyk) + M[9]) + M[13];
      const float wz = ((M[2] * cx + M[6] * cyk) + M[10]) + M[14];
      const float ww = ((M[3] * cx + M[7] * cyk) + M[11]) + M[15];
      corner[k][0] = wx / ww - cam[0];
      corner[k][1] = wy / ww - cam[1];
      corner[k][2] = wz / ww - cam[2];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float* u = corner[k];
      const float* w = corner[(k + 1) & 3];
      const float* in = corner[(k + 2) & 3];
      float nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
      const float len = sqrtf((nx * nx + ny * ny) + nz * nz);
      const float side = (nx * in[0] + ny * in[1]) + nz * in[2];
      const float scale = (len > 0.0f && len < 3.0e38f) ? ((side > 0.0f) ? -1.0f : 1.0f) / len : 0.0f;  // degenerate face: culls nothing
      face[k][0] = nx * scale;
      face[k][1] = ny * scale;
      face[k][2] = nz * scale;
    }
  }

  for (uint32_t base = 0; base < sc.sphere_count; base += kSphereChunk) {
    const uint32_t n = min((uint32_t)kSphereChunk, sc.sphere_count - base);
    __syncthreads();
    // Staged per sphere: camera - centre and |camera - centre|^2 - radius^2, the same for every pixel - but only
    // the spheres that a ray of this tile can hit, in table order (the nearest-hit rule below prefers the earlier
    // of two equal distances).  A ray's discriminant is b^2 - a*c = a * (r^2 - d^2) with d the distance from the
    // centre to the ray; in binary32 its sign is good to about 4e-7 * a * |camera - centre|^2, so a sphere can
    // (seem to) be hit only by rays within sqrt(r^2 + 1e-5 * |camera - centre|^2) of its centre: spheres farther
    // than that, plus a margin for the faces' own rounding, outside one of the four faces are left out.
    float4 rec = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool keep = false;
    if (t < n) {
      const vkr_sphere& sp = p.spheres[base + t];
      const float oc[3] = {cam[0] - sp.center[0], cam[1] - sp.center[1], cam[2] - sp.center[2]};
      const float oc2 = (oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2];
      rec = make_float4(oc[0], oc[1], oc[2], oc2 - sp.radius * sp.radius);
      const float reach = sqrtf(sp.radius * sp.radius + 1e-5f * oc2) + 1e-4f * sqrtf(oc2);
      keep = true;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float outside = -((face[k][0] * oc[0] + face[k][1] * oc[1]) + face[k][2] * oc[2]);  // distance of the centre beyond face k
        if (outside > reach) keep = false;  // false for NaN: kept
      }
    }
    const uint64_t kept = __ballot(keep);
    const uint32_t wv = t >> 6;
    if ((t & 63u) == 0u && wv < 2u) s_kept[wv] = (uint32_t)__popcll(kept);
    __syncthreads();
    if (keep) {
      const uint32_t at = (wv == 1u ? s_kept[0] : 0u) + __builtin_amdgcn_mbcnt_hi((uint32_t)(kept >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)kept, 0u));
      s_sphere[at] = rec;
      s_index[at] = base + t;
    }
    const uint32_t n_kept = s_kept[0] + s_kept[1];
    __syncthreads();
    if (n_kept == 0u) continue;
    float4 ahead = s_sphere[0];
    for (uint32_t s = 0; s < n_kept; ++s) {
      const float4 sp = ahead;
      ahead = s_sphere[min(s + 1u, n_kept - 1u)];  // requested before this sphere's arithmetic
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float b = (sp.x * dir[i][0] + sp.y * dir[i][1]) + sp.z * dir[i][2];
        const float disc = b * b - a[i] * sp.w;
        if (disc >= 0.0f) {
          const float tt = (-b - sqrtf(disc)) / a[i];
          if (tt > 0.0f && tt < hit[i].t) {
            hit[i].t = tt;
            hit[i].what = (int32_t)s_index[s];
          }
        }
      }
    }
  }
  const float gn[3] = {sc.ground_plane[0], sc.ground_plane[1], sc.ground_plane[2]};
  const float gn2 = (gn[0] * gn[0] + gn[1] * gn[1]) + gn[2] * gn[2];
  if (gn2 > 0.0f) {
    const float dist = ((gn[0] * cam[0] + gn[1] * cam[1]) + gn[2] * cam[2]) + sc.ground_plane[3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float denom = (gn[0] * dir[i][0] + gn[1] * dir[i][1]) + gn[2] * dir[i][2];
      if (denom < 0.0f) {  // the plane faces the ray
        const float tt = -dist / denom;
        if (tt > 0.0f && tt < hit[i].t) {
          hit[i].t = tt;
          hit[i].what = -2;
        }
      }
    }
  }
  if (!inside) return;

  const float* VP = sc.view_proj;
  uint32_t alb[4], mat[4];
  uint2 nrm[4];
  float dep[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (hit[i].what == -1) {  // the clear values of /root/reference/src/main.cpp:125-129
      alb[i] = 0xFF000000u;
      mat[i] = 0xFF000000u;
      nrm[i] = pack_half4(0.0f, 0.0f, 0.0f, 1.0f);
      dep[i] = 1.0f;
      continue;
    }
    const float P[3] = {cam[0] + dir[i][0] * hit[i].t, cam[1] + dir[i][1] * hit[i].t, cam[2] + dir[i][2] * hit[i].t};
    float N[3], col[3], rough, metal, occ;
    if (hit[i].what >= 0) {
      const vkr_sphere& s = p.spheres[hit[i].what];
      N[0] = (P[0] - s.center[0]) / s.radius;
      N[1] = (P[1] - s.center[1]) / s.radius;
      N[2] = (P[2] - s.center[2]) / s.radius;
      col[0] = s.albedo[0];
      col[1] = s.albedo[1];
      col[2] = s.albedo[2];
      rough = s.roughness;
      metal = s.metalness;
      occ = s.occlusion;
    } else {
      const float inv = 1.0f / sqrtf(gn2);
      N[0] = gn[0] * inv;
      N[1] = gn[1] * inv;
      N[2] = gn[2] * inv;
      const float s = floorf(P[0] / sc.ground_tile) + floorf(P[2] / sc.ground_tile);
      const bool odd = (s - 2.0f * floorf(s * 0.5f)) != 0.0f;
      const float* g = odd ? sc.ground_albedo_b : sc.ground_albedo_a;
      col[0] = g[0];
      col[1] = g[1];
      col[2] = g[2];
      rough = sc.ground_roughness;
      metal = sc.ground_metalness;
      occ = 1.0f;
    }
    const float cz = ((VP[2] * P[0] + VP[6] * P[1]) + VP[10] * P[2]) + VP[14];
    const float cw = ((VP[3] * P[0] + VP[7] * P[1]) + VP[11] * P[2]) + VP[15];
    alb[i] = pack_unorm8x4(col[0], col[1], col[2], 1.0f);
    mat[i] = pack_unorm8x4(occ, rough, metal, 1.0f);
    nrm[i] = pack_half4(-N[0], -N[1], -N[2], 1.0f);
    dep[i] = cz / cw;
  }
  const uint32_t valid = min(4u, p.o.width - x0);
  uint32_t* pa = reinterpret_cast<uint32_t*>(p.o.albedo + (size_t)y * p.o.pitch_albedo) + x0;
  uint32_t* pm = reinterpret_cast<uint32_t*>(p.o.material + (size_t)y * p.o.pitch_material) + x0;
  uint2* pn = reinterpret_cast<uint2*>(p.o.normal + (size_t)y * p.o.pitch_normal) + x0;
  float* pd = reinterpret_cast<float*>(p.o.depth + (size_t)y * p.o.pitch_depth) + x0;
  if (valid == 4u) {
    *reinterpret_cast<uint4*>(pa) = make_uint4(alb[0], alb[1], alb[2], alb[3]);
    *reinterpret_cast<uint4*>(pm) = make_uint4(mat[0], mat[1], mat[2], mat[3]);
    *reinterpret_cast<uint4*>(pn) = make_uint4(nrm[0].x, nrm[0].y, nrm[1].x, nrm[1].y);
    *reinterpret_cast<uint4*>(pn + 2) = make_uint4(nrm[2].x, nrm[2].y, nrm[3].x, nrm[3].y);
    *reinterpret_cast<float4*>(pd) = make_float4(dep[0], dep[1], dep[2], dep[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if ((uint32_t)i < valid) {
        pa[i] = alb[i];
        pm[i] = mat[i];
        pn[i] = nrm[i];
        pd[i] = dep[i];
      }
    }
  }
}

int32_t check_plane(const vkr_image2d& im, const char* what, uint32_t fmt_a, uint32_t fmt_b, uint32_t texel, uint32_t w, uint32_t h) {
  if (!im.device) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "%s: null image", what);
  if (im.format != fmt_a && im.format != fmt_b) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "%s: format %u not accepted", what, im.format);
  if (im.width != w || im.height != h) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "%s: extent %ux%u differs from %ux%u", what, im.width, im.height, w, h);
  if (im.pitch_bytes < w * texel || (im.pitch_bytes & 15u) || (reinterpret_cast<uintptr_t>(im.device) & 15u))
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "%s: pitch %u / base must be 16-byte aligned and cover a row", what, im.pitch_bytes);
  return VKR_OK;
}

int32_t planes_from(const vkr_image2d& albedo, const vkr_image2d& material, const vkr_image2d& normal, const vkr_image2d& depth, Planes* o) {
  const uint32_t w = albedo.width, h = albedo.height;
  if (w == 0u || h == 0u || w > 32768u || h > 32768u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "extent %ux%u out of range", w, h);
  if (int32_t rc = check_plane(albedo, "albedo", VKR_FORMAT_R8G8B8A8_UNORM, VKR_FORMAT_R8G8B8A8_UNORM, 4, w, h)) return rc;
  if (int32_t rc = check_plane(material, "material", VKR_FORMAT_R8G8B8A8_UNORM, VKR_FORMAT_R8G8B8A8_UNORM, 4, w, h)) return rc;
  if (int32_t rc = check_plane(normal, "normal", VKR_FORMAT_R16G16B16A16_SFLOAT, VKR_FORMAT_R16G16B16A16_SFLOAT, 8, w, h)) return rc;
  if (int32_t rc = check_plane(depth, "depth", VKR_FORMAT_D32_SFLOAT, VKR_FORMAT_R32_SFLOAT, 4, w, h)) return rc;
  o->albedo = static_cast<uint8_t*>(albedo.device);
  o->material = static_cast<uint8_t*>(material.device);
  o->normal = static_cast<uint8_t*>(normal.device);
  o->depth = static_cast<uint8_t*>(depth.device);
  o->pitch_albedo = albedo.pitch_bytes;
  o->pitch_material = material.pitch_bytes;
  o->pitch_normal = normal.pitch_bytes;
  o->pitch_depth = depth.pitch_bytes;
  o->width = w;
  o->height = h;
  return VKR_OK;
}

}  // namespace

extern "C" {

int32_t vkr_gbuffer_encode(vkr_context* ctx, const vkr_gbuffer_encode_args* a) {
  if (!ctx || !a) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode: null argument");
  Planes o;
  if (int32_t rc = planes_from(a->albedo, a->material, a->normal, a->depth, &o)) return rc;
  if (!a->albedo_f32 || !a->material_f32 || !a->normal_f32 || !a->depth_f32) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode: null source plane");
  if ((reinterpret_cast<uintptr_t>(a->albedo_f32) | reinterpret_cast<uintptr_t>(a->material_f32) | reinterpret_cast<uintptr_t>(a->normal_f32)) & 15u)
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode: float4 source planes must be 16-byte aligned");
  VKR_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(gbuffer_encode_kernel, dim3((o.width + 1023u) / 1024u, o.height), dim3(256), 0, ctx->stream,
                     static_cast<const float4*>(a->albedo_f32), static_cast<const float4*>(a->material_f32), static_cast<const float4*>(a->normal_f32),
                     static_cast<const float*>(a->depth_f32), o);
  VKR_HIP(hipGetLastError());
  return VKR_OK;
}

int32_t vkr_gbuffer_encode_surfaces(vkr_context* ctx, const vkr_gbuffer_surface_args* a) {
  if (!ctx || !a) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode_surfaces: null argument");
  Planes o;
  if (int32_t rc = planes_from(a->albedo, a->material, a->normal, a->depth, &o)) return rc;
  const void* sources[] = {a->albedo_texel, a->specular_texel, a->normal_texel, a->normal_obj, a->tangent_obj, a->model_matrices};
  for (const void* src : sources) {
    if (!src) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode_surfaces: null source plane or matrix table");
    if (reinterpret_cast<uintptr_t>(src) & 15u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode_surfaces: float4 sources and matrices must be 16-byte aligned");
  }
  if (!a->depth_f32) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode_surfaces: null depth plane");
  if (a->matrix_count == 0u) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_encode_surfaces: empty matrix table");
  SurfaceSources s;
  s.albedo = static_cast<const float4*>(a->albedo_texel);
  s.specular = static_cast<const float4*>(a->specular_texel);
  s.normal_texel = static_cast<const float4*>(a->normal_texel);
  s.normal_obj = static_cast<const float4*>(a->normal_obj);
  s.tangent_obj = static_cast<const float4*>(a->tangent_obj);
  s.depth = static_cast<const float*>(a->depth_f32);
  s.matrices = static_cast<const float4*>(a->model_matrices);
  s.matrix_count = a->matrix_count;
  VKR_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(gbuffer_surface_kernel, dim3((o.width + 511u) / 512u, o.height), dim3(256), 0, ctx->stream, s, o);
  VKR_HIP(hipGetLastError());
  return VKR_OK;
}

int32_t vkr_gbuffer_fill(vkr_context* ctx, const vkr_gbuffer_fill_args* a) {
  if (!ctx || !a) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_fill: null argument");
  FillParams p{};
  if (int32_t rc = planes_from(a->albedo, a->material, a->normal, a->depth, &p.o)) return rc;
  if (a->scene.sphere_count && !a->spheres) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_fill: %u spheres but a null table", a->scene.sphere_count);
  if (!(a->scene.ground_tile > 0.0f)) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_fill: ground_tile must be > 0");
  p.spheres = static_cast<const vkr_sphere*>(a->spheres);
  p.row_offset = a->row_offset;
  p.full_height = a->full_height ? a->full_height : p.o.height;
  if ((uint64_t)p.row_offset + p.o.height > p.full_height)
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_gbuffer_fill: rows [%u, %u) exceed full_height %u", p.row_offset, p.row_offset + p.o.height, p.full_height);
  p.scene = a->scene;
  VKR_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(gbuffer_fill_kernel, dim3((p.o.width + 31u) / 32u, (p.o.height + 31u) / 32u), dim3(256), 0, ctx->stream, p);
  VKR_HIP(hipGetLastError());
  return VKR_OK;
}

}  // extern "C"
