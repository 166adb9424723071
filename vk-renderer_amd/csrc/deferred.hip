// Deferred lighting resolve for gfx950 (MI355X): the pass of
// /root/reference/shaders/deferred.comp:98-131, hand-written for wave64.
//
// This file is compiled with -ffp-contract=off: every fused multiply-add below is an explicit
// __builtin_fmaf, so the depth -> world-position step can reproduce the reference's binary32
// evaluation order bit for bit (it is ill-conditioned: w = a*z + b cancels as z -> 1), while the
// per-light loop is written in a cancellation-free form that needs 4 transcendental and 60 plain
// VALU instructions per (pixel, light).
//
// Per-light algebra (unit vectors v, l, n; c = l.n; vn = v.n; H = v + l un-normalised, split into
// its component along n, Hn = vn + c, and its tangential part Ht; HH = |H|^2 = Ht^2 + Hn^2):
//   h.n^2            = Hn^2 / HH
//   (a2-1) h.n^2 + 1 = (a2*Hn^2 + Ht^2) / HH            =: dn / HH      (no cancellation)
//   D   = a2 / (PI d^2)            = a2 * HH^2 / (PI * dn^2)            (deferred.comp:59-62)
//   v.h = l.h = sqrt(HH) / 2,  (v.h)^2 = HH / 4                          (deferred.comp:47,89)
//   f90 = 0.5 r + 2 (l.h)^2 r      = 0.5 r + 0.5 r HH                    (deferred.comp:48-50)
//   Lambda(c) + 1 = (c + sqrt(a2 c^2 + 1)) / (2c)                        (deferred.comp:64-70)
//   G   = 1 / ((Lambda(vn)+1)(Lambda(c)+1))                              (deferred.comp:72-76)
// so  specular = F * D * G * 0.25 / vn = F * [HH^2 * c / (dn^2 (c + q))] * Kspec,
//     q = sqrt(a2 c^2 + 1),  Kspec = a2 / (PI (vn + sqrt(a2 vn^2 + 1)))  (per pixel)
// and F = Cs + (1 - Cs) (1 - v.h)^5 splits the light sum into two colour-weighted sums.
// The light vector is rotated into a per-pixel orthonormal frame (T, B, N) so that Ht is obtained
// directly as (l.T + v.T, l.B + v.B) instead of as |H|^2 - Hn^2.
//
// Instruction selection follows what the gfx950 vector unit was measured to do (4 waves per SIMD,
// tools/ubench_valu2.hip): mul/add/fma with VGPR, inline-constant or literal operands and the
// clamp output modifier issue every 2 cycles; anything with an SGPR operand, v_min/v_max/v_med3,
// compares and conversions take ~3.2; a lone transcendental in a plain stream costs 13-17 cycles,
// but four adjacent ones cost ~2.8 each.  Hence: light records are staged in LDS and read back
// with broadcast ds_read_b128 into VGPRs (free next to VALU work) instead of being used as SGPR
// operands; min/max/select are replaced by clamp-modifier arithmetic; and four (pixel, light)
// pairs advance in lockstep.  A transcendental costs as much as ~8 plain instructions in this mix
// (measured by ablation on this kernel: 2.07 ms with the 16 per iteration replaced by moves, 2.87 ms
// with them); a degree-5 polynomial for sqrt(1 + a2 c^2) measured the same as the instruction, so
// the instruction (1 ulp) stays.  Where the four sit in the stream (grouped as here, left to the
// compiler, or spread one per ~14 plain instructions with sched_group_barrier) moved the kernel by
// less than +-4 %.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "vkr_kernels.h"

#ifndef VKR_TILE_BATCH_FIRST
#define VKR_TILE_BATCH_FIRST 8u
#endif

namespace {

constexpr float kPi = 3.14159f;  // deferred.comp:3 (truncated on purpose)
constexpr float kAmbient = 0.05f;
constexpr float kAlphaFloor = 0.0064f;
constexpr float kEarlyOut = 1e-6f;
constexpr float kDistFloor = 1e-7f;        // deferred.comp:122
constexpr float kInvDistFloor = 1e7f;      // 1 / kDistFloor
constexpr float kStepScale = 0x1p100f;     // (c - 1e-6) * 2^100 saturates the clamp for any c > 1e-6
constexpr float kRadiusStepScale = 0x1p80f;
constexpr int kChunk = 512;                // lights per LDS chunk: 2 float4 each = 16 KiB
constexpr int kTileSpan = 4096;            // tile light lists: lights tested per pass (64 groups of 64)

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float clamp01_(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
__device__ __forceinline__ float pow5_(float t) {
  const float t2 = t * t;
  const float t4 = t2 * t2;
  return t4 * t;
}
__device__ __forceinline__ float const_inv151() { return __uint_as_float(0x3F298953u); }  // 1.0/1.51 as folded

// Four transcendentals, each followed by one idle cycle (s_nop 0), issued under raised wave priority.
// Measured on gfx950 at 4 waves per SIMD (tools/gen_ubench_trans3.py, gen_ubench_trans4.py), in a stream
// of plain VALU work: a transcendental whose wave tries to issue its next VALU instruction in the very
// next cycle costs 14 plain issue slots; with one idle cycle after it, 6.4; and 4.8 when the four issue
// back to back without another wave's instructions in between, which the priority bracket arranges.
// The idle cycle also covers the wait state a transcendental result needs before a plain VALU
// instruction may read it, which the compiler cannot see inside an asm statement.
#define VKR_TRANS4(NAME, OP)                                                                            \
  __device__ __forceinline__ void NAME(float (&o)[4], const float (&x)[4]) {                            \
    asm("s_setprio 3\n\t" OP " %0, %4\n\ts_nop 0\n\t" OP " %1, %5\n\ts_nop 0\n\t" OP " %2, %6\n\ts_nop 0\n\t" OP \
        " %3, %7\n\ts_nop 0\n\ts_setprio 0"                                                          \
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3])                                            \
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));                                                  \
  }
VKR_TRANS4(rsq4, "v_rsq_f32")
VKR_TRANS4(sqrt4, "v_sqrt_f32")
VKR_TRANS4(rcp4, "v_rcp_f32")
#undef VKR_TRANS4

// Correctly rounded a/b for operands whose quotient and reciprocal stay in the normal range (the
// compiler's own division is this same arithmetic wrapped in v_div_scale / v_div_fmas / v_div_fixup,
// which only matter for denormal, huge or special operands and issue at half rate or worse).
// The refined reciprocal is shared by every division by the same b.
struct ExactRcp {
  float b, r;
};
__device__ __forceinline__ ExactRcp exact_rcp(float b) {
  float r = rcp_(b);
  r = fma_(fma_(-b, r, 1.0f), r, r);
  return {b, r};
}
__device__ __forceinline__ float exact_div(float a, const ExactRcp& d) {
  float q = a * d.r;
  q = fma_(fma_(-d.b, q, a), d.r, q);
  q = fma_(fma_(-d.b, q, a), d.r, q);
  return q;
}

struct Pixel {
  // geometry
  float px, py, pz;            // world position (bit-identical to the binary32 reference order)
  float tx, ty, tz;            // tangent frame
  float bx, by, bz;
  float nx, ny, nz;            // shading normal = -normalize(gbuffer normal)
  float vt, vb, vn;            // view direction in the frame
  // material
  float a2, rh, ebm1, pv;      // alpha^2, roughness/2, energyBias-1, (1-vn)^5
  // accumulators: 1e-7 * sum_j color_j * {diffuse, spec, spec*(1-v.h)^5} scalar weights
  float ad[3], as0[3], as1[3];
  float in_range;              // radius cutoff only: > 0 once any light lay inside the radius
};

struct Light {
  float px, py, pz, cx, cy, cz;
};

// Two lights applied to two pixels: the four (pixel, light) pairs k = 0..3 are (a,A) (b,A) (a,B) (b,B).
// Every weight carries a factor 1e-7 (the distance floor is applied as clamp(inv^2 * 1e-7), which
// equals min(1/d^2, 1e7) * 1e-7); finish_pixel() multiplies it back.
struct LoopConsts {
  float step_scale, step_bias, dist_floor, radius2_scaled;
};

template <bool kRadius>
__device__ __forceinline__ void shade4(Pixel& a, Pixel& b, const Light& A, const Light& B, const LoopConsts& kc) {
  float dll[4], lt[4], lb[4], ln[4], inv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Pixel& s = (k & 1) ? b : a;
    const Light& l = (k & 2) ? B : A;
    const float wx = l.px - s.px, wy = l.py - s.py, wz = l.pz - s.pz;
    dll[k] = fma_(wz, wz, fma_(wy, wy, wx * wx));
    lt[k] = fma_(wz, s.tz, fma_(wy, s.ty, wx * s.tx));
    lb[k] = fma_(wz, s.bz, fma_(wy, s.by, wx * s.bx));
    ln[k] = fma_(wz, s.nz, fma_(wy, s.ny, wx * s.nx));
  }
  rsq4(inv, dll);
  float c[4], hh[4], dn[4], att[4], sh[4], qa[4], q[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Pixel& s = (k & 1) ? b : a;
    c[k] = ln[k] * inv[k];
    const float ht = fma_(lt[k], inv[k], s.vt);
    const float hb = fma_(lb[k], inv[k], s.vb);
    const float hn = c[k] + s.vn;
    const float ht2 = fma_(hb, hb, ht * ht);
    const float hn2 = hn * hn;
    hh[k] = ht2 + hn2;
    dn[k] = fma_(s.a2, hn2, ht2);
    att[k] = clamp01_((inv[k] * inv[k]) * kc.dist_floor);  // min(1/d^2, 1e7) * 1e-7
  }
  sqrt4(sh, hh);
  float kd[4], pvh[4], den[4], rden[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Pixel& s = (k & 1) ? b : a;
    pvh[k] = pow5_(clamp01_(fma_(sh[k], -0.5f, 1.0f)));  // (1 - v.h)^5
    const float f90m1 = fma_(hh[k], s.rh, s.ebm1);
    const float pl = pow5_(clamp01_(1.0f - c[k]));
    kd[k] = fma_(f90m1, pl, 1.0f) * fma_(f90m1, s.pv, 1.0f);
    qa[k] = fma_(s.a2, c[k] * c[k], 1.0f);
  }
  sqrt4(q, qa);
#pragma unroll
  for (int k = 0; k < 4; ++k) den[k] = fma_(dn[k] * dn[k], c[k] + q[k], 1e-37f);  // +1e-37 keeps culled lanes finite
  rcp4(rden, den);
  float wd[4], ws[4], wp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // early-out of deferred.comp:43,84 as a multiplier: (c - 1e-6) * 2^100 is exact inside the FMA; its clamp
    // is 1 for every c > 1e-6 and 0 for c <= 1e-6 or NaN
    float ci = (c[k] * clamp01_(fma_(c[k], kc.step_scale, kc.step_bias))) * att[k];
    if (kRadius) {
      const float near = clamp01_(fma_(dll[k], -kRadiusStepScale, kc.radius2_scaled));  // 1 iff dist^2 < radius^2
      ci *= near;
      Pixel& s = (k & 1) ? b : a;
      s.in_range += near;
    }
    const float ks0 = (hh[k] * hh[k]) * rden[k];
    wd[k] = kd[k] * ci;
    ws[k] = ks0 * ci;
    wp[k] = ws[k] * pvh[k];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Pixel& s = (k & 1) ? b : a;
    const Light& l = (k & 2) ? B : A;
    s.ad[0] = fma_(l.cx, wd[k], s.ad[0]);
    s.ad[1] = fma_(l.cy, wd[k], s.ad[1]);
    s.ad[2] = fma_(l.cz, wd[k], s.ad[2]);
    s.as0[0] = fma_(l.cx, ws[k], s.as0[0]);
    s.as0[1] = fma_(l.cy, ws[k], s.as0[1]);
    s.as0[2] = fma_(l.cz, ws[k], s.as0[2]);
    s.as1[0] = fma_(l.cx, wp[k], s.as1[0]);
    s.as1[1] = fma_(l.cy, wp[k], s.as1[1]);
    s.as1[2] = fma_(l.cz, wp[k], s.as1[2]);
  }
}

// deferred.comp:99-100: clip-space coordinate of a pixel centre, true division included.
__device__ __forceinline__ float clip_coord(uint32_t i, const ExactRcp& by_extent) {
  return exact_div((float)i + 0.5f, by_extent) * 2.0f - 1.0f;
}

__device__ __forceinline__ float unorm8(uint32_t word, int byte) { return (float)((word >> (8 * byte)) & 0xFFu) * (1.0f / 255.0f); }

// deferred.comp:99-118: everything that does not depend on the light.
__device__ __forceinline__ void setup_pixel(Pixel& s, const DeferredParams& p, float cx, float cy, float depth, uint32_t material,
                                            uint32_t nrm_lo, uint32_t nrm_hi) {
  // :99-103 in the reference's binary32 order, true divisions included (no contraction in this file).
  // Position reconstruction is ill-conditioned twice over: the w row cancels as depth -> 1, and a light
  // 0.4 units from a surface 400 units out turns one ulp of position into 1.5e-4 of 1/d^2.  So the
  // position is reproduced bit for bit rather than to within rounding.
  const float* M = p.inv_vp;
  const float wx = ((M[0] * cx + M[4] * cy) + M[8] * depth) + M[12];
  const float wy = ((M[1] * cx + M[5] * cy) + M[9] * depth) + M[13];
  const float wz = ((M[2] * cx + M[6] * cy) + M[10] * depth) + M[14];
  const float ww = ((M[3] * cx + M[7] * cy) + M[11] * depth) + M[15];
  const ExactRcp by_w = exact_rcp(ww);
  s.px = exact_div(wx, by_w);
  s.py = exact_div(wy, by_w);
  s.pz = exact_div(wz, by_w);

  const float rough = unorm8(material, 1);

  // :108 normal = -normalize(n)
  const __half2 h01 = __builtin_bit_cast(__half2, nrm_lo);
  const __half2 h23 = __builtin_bit_cast(__half2, nrm_hi);
  const float rx = __low2float(h01), ry = __high2float(h01), rz = __low2float(h23);
  const float rinv = -rsq_(fma_(rz, rz, fma_(ry, ry, rx * rx)));
  s.nx = rx * rinv;
  s.ny = ry * rinv;
  s.nz = rz * rinv;

  // orthonormal frame around N (branch-free construction)
  const float sg = __builtin_copysignf(1.0f, s.nz);
  const float a = -rcp_(sg + s.nz);
  const float b = s.nx * s.ny * a;
  s.tx = fma_(sg * s.nx, s.nx * a, 1.0f);
  s.ty = sg * b;
  s.tz = -sg * s.nx;
  s.bx = b;
  s.by = fma_(s.ny, s.ny * a, sg);
  s.bz = -s.ny;

  // :118 viewDir = normalize(cameraPos - fragPos)
  const float ex = p.cam[0] - s.px, ey = p.cam[1] - s.py, ez = p.cam[2] - s.pz;
  const float einv = rsq_(fma_(ez, ez, fma_(ey, ey, ex * ex)));
  const float vx = ex * einv, vy = ey * einv, vz = ez * einv;
  s.vt = fma_(vz, s.tz, fma_(vy, s.ty, vx * s.tx));
  s.vb = fma_(vz, s.bz, fma_(vy, s.by, vx * s.bx));
  s.vn = fma_(vz, s.nz, fma_(vy, s.ny, vx * s.nx));

  // :116 alpha = max(r*r, 0.0064)
  const float alpha = fmaxf(rough * rough, kAlphaFloor);
  s.a2 = alpha * alpha;
  s.rh = 0.5f * rough;
  s.ebm1 = fma_(0.5f, rough, -1.0f);
  s.pv = pow5_(clamp01_(1.0f - s.vn));
#pragma unroll
  for (int k = 0; k < 3; ++k) s.ad[k] = s.as0[k] = s.as1[k] = 0.0f;
  s.in_range = 0.0f;
}

// Combines the three light sums into the pixel colour (deferred.comp:107,114-115,127-128).  The
// albedo and material words are decoded here, after the light loop, so that nothing of them occupies
// registers during it.
// `lit_by_any`: at least one light took part for this pixel (any light at all, or with the radius
// cutoff one inside the radius).  A pixel whose normal or position is not a number (a cleared G-buffer
// texel has normal 0: normalize(0)) then comes out NaN, as every term of deferred.comp:120-129 does
// (rgba8 stores it as 0); with no light taking part it keeps its ambient colour.
__device__ __forceinline__ void finish_pixel(const Pixel& s, uint32_t albedo, uint32_t material, bool lit_by_any, float out[3]) {
  const float rough = unorm8(material, 1), metal = unorm8(material, 2);
  const bool front = s.vn > kEarlyOut;  // deferred.comp:43,84 (vDotN half of the early-out)
  const float unlit = (lit_by_any && !(s.vn == s.vn)) ? __builtin_nanf("") : 0.0f;
  const float ef = fma_(const_inv151(), rough, 1.0f - rough);  // mix(1, 1/1.51, r)
  const float kdiff = (kPi * kInvDistFloor) * ef;  // the light sums carry a factor 1e-7
  const float kspec = (s.a2 * kInvDistFloor) * rcp_(kPi * (s.vn + sqrt_(fma_(s.a2, s.vn * s.vn, 1.0f))));
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float alb = unorm8(albedo, k);
    const float cd = alb * (1.0f - metal);  // mix(albedo, 0, metal)
    const float cs = alb * metal;           // mix(0, albedo, metal)
    const float lit = fma_(cd * kdiff, s.ad[k], kspec * fma_(cs, s.as0[k], (1.0f - cs) * s.as1[k]));
    out[k] = fma_(kAmbient, alb, front ? lit : unlit);
  }
}

__device__ __forceinline__ uint32_t quantize_unorm8(float c) {
  // NaN -> 0, clamp, round-half-even (Vulkan float -> unorm conversion of the rgba8 store, :23,130)
  const float t = (c > 0.0f) ? fminf(c, 1.0f) : 0.0f;
  return (uint32_t)__builtin_rintf(t * 255.0f);
}

// Work decomposition: a 256-thread workgroup shades a 32 x 32 pixel tile; lane t owns the four
// consecutive pixels (4*(t&7) .. +3, t>>3) of the tile, so every G-buffer plane is read with one
// 128-bit load per lane (two for the 8-byte normals) and the 8-bit result is written with one.
// The lane shades its four pixels as two pairs; each pair runs the light loop two lights at a time.
// The light records (pos.xyz, color.xyz as two float4) go through LDS in lists of up to kChunk,
// padded to an even count with a black light:
//   kTile = false : the whole table, chunk after chunk;
//   kTile = true  : (radius cutoff only) the lights whose sphere of influence reaches the tile's
//                   world-space bounding box, in table order.
// launch bound: 4 waves per SIMD (<= 128 VGPRs); measured 1-2 % faster than the 3 waves the allocator
// settles on unprompted.  The tile-list variants get 3 waves (<= 168 VGPRs): at 128 they spill some 30
// registers, and a kernel that uses scratch memory at all measured 0.14 ms slower per 8K frame.
template <int kOutMode, bool kRadius, bool kTile>
__global__ __launch_bounds__(256, kTile ? 3 : 4) void deferred_kernel(const DeferredParams p) {
  __shared__ float4 s_light[2 * kChunk + 4];  // +1 pair: the prefetch of the last iteration reads past the list
  __shared__ uint32_t s_stash[14][256];  // rows: see kRow* below
  __shared__ float s_red[8];
  const uint32_t t = threadIdx.x;
  const uint32_t x0 = blockIdx.x * 32u + (t & 7u) * 4u;
  const uint32_t y = blockIdx.y * 32u + (t >> 3);
  const bool inside = (x0 < p.width) && (y < p.height);
  // lanes outside the image shade the tile's first group (in bounds by construction) and store nothing
  const uint32_t lx = inside ? x0 : blockIdx.x * 32u;
  const uint32_t ly = inside ? y : blockIdx.y * 32u;
  const uint32_t y_full = ly + p.row_offset;

  const uint4 alb4 = *reinterpret_cast<const uint4*>(p.albedo + (size_t)ly * p.pitch_albedo + (size_t)lx * 4u);
  const uint4 mat4 = *reinterpret_cast<const uint4*>(p.material + (size_t)ly * p.pitch_material + (size_t)lx * 4u);
  const float4 dep4 = *reinterpret_cast<const float4*>(p.depth + (size_t)ly * p.pitch_depth + (size_t)lx * 4u);
  const uint4 nrmA = *reinterpret_cast<const uint4*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)lx * 8u);
  const uint4 nrmB = *reinterpret_cast<const uint4*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)lx * 8u + 16u);

  // Little more than what the light loop itself reads stays in registers across it.  The albedo and
  // material words of all four pixels (decoded after the loop), the normal words of the second pixel
  // pair and the first pair's results wait in LDS, one dword column per lane (conflict-free, private to
  // the lane: no barrier needed).  Left in VGPRs, the register allocator spills part of them to scratch
  // memory, which is HBM traffic of the order of the output itself.  The rows are kept few: past
  // 32 KiB of LDS per workgroup the kernel measurably slows down.
  enum { kRowAlbedo = 0, kRowMaterial = 4, kRowNormal = 8, kRowExtra = 12, kStashRows = 14 };
  const ExactRcp by_width = exact_rcp((float)p.width), by_height = exact_rcp((float)p.full_height);
  const float cy = clip_coord(y_full, by_height);
  float cx[2] = {clip_coord(lx, by_width), clip_coord(lx + 1u, by_width)};
  const float cx_next[2] = {clip_coord(lx + 2u, by_width), clip_coord(lx + 3u, by_width)};
  float dep[2] = {dep4.x, dep4.y};
  uint32_t mat[2] = {mat4.x, mat4.y};
  uint32_t nlo[2] = {nrmA.x, nrmA.z};
  uint32_t nhi[2] = {nrmA.y, nrmA.w};
  s_stash[kRowAlbedo + 0][t] = alb4.x;
  s_stash[kRowAlbedo + 1][t] = alb4.y;
  s_stash[kRowAlbedo + 2][t] = alb4.z;
  s_stash[kRowAlbedo + 3][t] = alb4.w;
  s_stash[kRowMaterial + 0][t] = mat4.x;
  s_stash[kRowMaterial + 1][t] = mat4.y;
  s_stash[kRowMaterial + 2][t] = mat4.z;
  s_stash[kRowMaterial + 3][t] = mat4.w;
  s_stash[kRowNormal + 0][t] = nrmB.x;
  s_stash[kRowNormal + 1][t] = nrmB.y;
  s_stash[kRowNormal + 2][t] = nrmB.z;
  s_stash[kRowNormal + 3][t] = nrmB.w;
  float dmin_lane = fminf(fminf(dep4.x, dep4.y), fminf(dep4.z, dep4.w));
  float dmax_lane = fmaxf(fmaxf(dep4.x, dep4.y), fmaxf(dep4.z, dep4.w));

  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  const uint32_t L = p.light_count;
  LoopConsts kc;
  // loop constants pinned in VGPRs: as SGPR or literal operands they would halve the issue rate of their users
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.dist_floor) : "s"(kDistFloor));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.radius2_scaled) : "s"(p.radius2 * kRadiusStepScale));
  const bool wave_live = __ballot(inside) != 0ull;  // wave-uniform
  const uint32_t lane = t & 63u, wv = t >> 6;

  // ---- tile bounds (kTile): world-space box around every pixel position of the tile ----------------
  // s_tile: box_lo[3], box_hi[3], four side faces of the tile's pyramid (outward unit normal + offset),
  // reach = radius + reconstruction pad.  Block-uniform, so it lives in LDS (broadcast reads) rather
  // than in 23 VGPRs of every lane.
  __shared__ float s_tile[23];
  __shared__ uint64_t s_group_hits[kTileSpan / 64];        // kTile: ballot word of each 64-light group of the span
  __shared__ uint32_t s_group_before[kTileSpan / 64 + 1];  // kTile: listed lights before each group
  if (kTile) {
    float dmin = dmin_lane, dmax = dmax_lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      dmin = fminf(dmin, __shfl_xor(dmin, off));
      dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    }
    if (lane == 0u) {
      s_red[wv] = dmin;
      s_red[4u + wv] = dmax;
    }
    __syncthreads();
    if (wv == 0u) {  // block-uniform values: one wave computes them, the others wait at the barrier below
      dmin = fminf(fminf(s_red[0], s_red[1]), fminf(s_red[2], s_red[3]));
      dmax = fmaxf(fmaxf(s_red[4], s_red[5]), fmaxf(s_red[6], s_red[7]));
      // The pixel positions are the images of points of the clip-space box [xa,xb] x [ya,yb] x [dmin,dmax]
      // under the projective map inverseVP; with w > 0 on all eight corners their hull contains them all.
      const float xa = (float)(blockIdx.x * 32u) / (float)p.width * 2.0f - 1.0f;
      const float xb = (float)min(blockIdx.x * 32u + 32u, p.width) / (float)p.width * 2.0f - 1.0f;
      const float ya = (float)(blockIdx.y * 32u + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
      const float yb = (float)(min(blockIdx.y * 32u + 32u, p.height) + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
      // Twelve points, one per lane: lane 4*kind + c holds corner c (xa ya, xb ya, xb yb, xa yb) at depth dmin
      // (kind 0, the near points), dmax (kind 1, the far points) or dmin - 0.25 (kind 2, a second point of the
      // same pixel ray).  Lanes from 12 up repeat them.
      const uint32_t c = lane & 3u, kind = (lane >> 2) % 3u;
      const float pcx = (c == 1u || c == 2u) ? xb : xa;
      const float pcy = (c >= 2u) ? yb : ya;
      const float pcz = kind == 0u ? dmin : (kind == 1u ? dmax : dmin - 0.25f);
      const float* M = p.inv_vp;
      const float wx = ((M[0] * pcx + M[4] * pcy) + M[8] * pcz) + M[12];
      const float wy = ((M[1] * pcx + M[5] * pcy) + M[9] * pcz) + M[13];
      const float wz = ((M[2] * pcx + M[6] * pcy) + M[10] * pcz) + M[14];
      const float ww = ((M[3] * pcx + M[7] * pcy) + M[11] * pcz) + M[15];
      const float q[3] = {wx / ww, wy / ww, wz / ww};
      const bool ok = (ww > 1e-30f) && (fabsf(q[0]) < 1e30f) && (fabsf(q[1]) < 1e30f) && (fabsf(q[2]) < 1e30f);
      const bool sane = (__ballot(ok) & 0xFFFull) == 0xFFFull;
      // box over the near and far points: lanes 0..7, which the xor offsets 1, 2, 4 keep among themselves
      float box_lo[3], box_hi[3];
      float scale = 0.0f;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        box_lo[a] = q[a];
        box_hi[a] = q[a];
#pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {
          box_lo[a] = fminf(box_lo[a], __shfl_xor(box_lo[a], off));
          box_hi[a] = fmaxf(box_hi[a], __shfl_xor(box_hi[a], off));
        }
        scale = fmaxf(scale, fmaxf(fabsf(box_lo[a]), fabsf(box_hi[a])));
      }
      // binary32 reconstruction error of a position is ~2e-4 relative at the far end of the depth range
      const float pad = 2e-3f * scale + 1e-2f;
      // the four side faces of the tile's pyramid, face c in lane c: plane through rays c and c+1, normal pointing
      // away from ray c+2
      float near1[3], near2[3], ray[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        near1[a] = __shfl(q[a], (int)((lane + 1u) & 3u));
        near2[a] = __shfl(q[a], (int)((lane + 2u) & 3u));
        ray[a] = __shfl(q[a], (int)(8u + c));
      }
      const float ux = ray[0] - q[0], uy = ray[1] - q[1], uz = ray[2] - q[2];
      const float vx = near1[0] - q[0], vy = near1[1] - q[1], vz = near1[2] - q[2];
      float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
      const float len2 = nx * nx + ny * ny + nz * nz;
      const float inv = len2 > 0.0f ? rsq_(len2) : 0.0f;  // degenerate face: zero normal, never culls
      nx *= inv; ny *= inv; nz *= inv;
      const float side = nx * (near2[0] - q[0]) + ny * (near2[1] - q[1]) + nz * (near2[2] - q[2]);
      const float flip = side > 0.0f ? -1.0f : 1.0f;
      const float fx = sane ? flip * nx : 0.0f, fy = sane ? flip * ny : 0.0f, fz = sane ? flip * nz : 0.0f;
      if (lane < 4u) {
        s_tile[6u + 4u * lane + 0u] = fx;
        s_tile[6u + 4u * lane + 1u] = fy;
        s_tile[6u + 4u * lane + 2u] = fz;
        // signed distance of x to the face = dot(n, x) - dot(n, near point c)
        s_tile[6u + 4u * lane + 3u] = -(fx * q[0] + fy * q[1] + fz * q[2]);
      }
      if (lane == 0u) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          s_tile[a] = sane ? box_lo[a] - pad : -INFINITY;
          s_tile[3 + a] = sane ? box_hi[a] + pad : INFINITY;
        }
        s_tile[22] = sqrt_(p.radius2) + pad;  // reach
      }
    }
    __syncthreads();
  }

  // kTile: the candidates of this tile are the lights of its coarse bin when the launch carries bins, else
  // the whole table; positions below index that candidate list.
  const uint32_t bin_index = (blockIdx.y * 32u / kBinPixels) * p.bins_x + blockIdx.x * 32u / kBinPixels;
  const uint32_t* __restrict__ bin = (kTile && p.bin_list) ? p.bin_list + (size_t)bin_index * L : nullptr;
  const uint32_t LL = bin ? p.bin_count[bin_index] : L;

  // kTile, pass 1 over the candidates [span, span + kTileSpan): every wave tests its share of the span's 64-light
  // groups against the tile bounds and leaves one ballot word per group in LDS - no barrier between groups -
  // then one wave turns the popcounts into list offsets.
  auto test_span = [&](uint32_t span, auto batch) {
    constexpr uint32_t kBatch = decltype(batch)::value;  // groups a wave tests per trip: that many position loads in flight
    const uint32_t span_lights = min((uint32_t)kTileSpan, LL - span);
    const uint32_t groups = (span_lights + 63u) >> 6;
    const float reach = s_tile[22];
    for (uint32_t grp0 = wv; grp0 < groups; grp0 += 4u * kBatch) {
      float4 lp[kBatch];
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t pos = min(span + (grp0 + 4u * u) * 64u + lane, LL - 1u);
        lp[u] = lights[3u * (bin ? bin[pos] : pos)];
      }
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t grp = grp0 + 4u * u;
        const uint32_t pos = span + grp * 64u + lane;
        const float dx = fmaxf(fmaxf(s_tile[0] - lp[u].x, lp[u].x - s_tile[3]), 0.0f);
        const float dy = fmaxf(fmaxf(s_tile[1] - lp[u].y, lp[u].y - s_tile[4]), 0.0f);
        const float dz = fmaxf(fmaxf(s_tile[2] - lp[u].z, lp[u].z - s_tile[5]), 0.0f);
        bool hit = (pos < LL) & ((dx * dx + dy * dy + dz * dz) < p.radius2);  // NaN position: never contributes, never listed
        if (__ballot(hit) != 0ull) {  // the box rejects most lights of most groups: one uniform branch skips the face tests
#pragma unroll
          for (int c = 0; c < 4; ++c) {  // outside a side face by more than the radius: cannot reach any pixel of the tile
            const float* f = &s_tile[6 + 4 * c];
            hit &= !(((f[0] * lp[u].x + f[1] * lp[u].y) + f[2] * lp[u].z) + f[3] > reach);  // no short circuit: straight-line code
          }
        }
        const uint64_t hits = __ballot(hit);
        if (lane == 0u && grp < groups) s_group_hits[grp] = hits;
      }
    }
    __syncthreads();
    if (wv == 0u) {
      const uint32_t mine = lane < groups ? (uint32_t)__popcll(s_group_hits[lane]) : 0u;
      uint32_t incl = mine;
#pragma unroll
      for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      s_group_before[lane + 1u] = incl;
      if (lane == 0u) s_group_before[0] = 0u;
    }
    __syncthreads();
  };
  // The first span is tested before any per-pixel state exists (fewer live registers); with at most
  // kTileSpan lights that is the only pass 1 of the workgroup, and both pixel pairs use its words.
  if (kTile && LL > 0u) test_span(0u, std::integral_constant<uint32_t, VKR_TILE_BATCH_FIRST>());

  bool list_complete = false;  // kTile: the list built for the first pixel pair holds every reaching light
  uint32_t list_count = 0;
  float last[2][3] = {};  // float outputs: the second pair's colours

  // results: the first pair's go to LDS (over the rows of the words just consumed) until the second
  // pair is done, so that each lane writes its four pixels with full-width stores
  auto keep_result = [&](int g, int i, const float (&rgb)[3]) {
    if (kOutMode == kOutRgba8) {
      s_stash[kRowAlbedo + g + i][t] = quantize_unorm8(rgb[0]) | (quantize_unorm8(rgb[1]) << 8) | (quantize_unorm8(rgb[2]) << 16) | 0xFF000000u;
    } else if (g == 0) {
      s_stash[kRowAlbedo + i][t] = __float_as_uint(rgb[0]);
      s_stash[kRowMaterial + i][t] = __float_as_uint(rgb[1]);
      s_stash[kRowExtra + i][t] = __float_as_uint(rgb[2]);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) last[i][k] = rgb[k];
    }
  };

  // kTile: no light reaches this tile (block-uniform; known after the only pass 1 when the candidates fit one
  // span).  Every light sum of deferred.comp:120-129 is then zero and the pixel is its ambient term
  // (deferred.comp:107), whatever its position and normal are: nothing of the per-pixel setup is needed.
  // (Only the tile-list variants take this path: as a run-time test in the other variants, for a table of
  // zero lights, it costs them four spilled registers.)
  const bool tile_dark = kTile && LL <= (uint32_t)kTileSpan && (LL == 0u || s_group_before[(LL + 63u) >> 6] == 0u);

  // One copy of the light loop serves both pixel pairs (no unrolling): half the code, same speed.
#pragma unroll 1
  for (int g = 0; g < 4; g += 2) {
    if (tile_dark) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint32_t albedo = s_stash[kRowAlbedo + g + i][t];
        const float rgb[3] = {kAmbient * unorm8(albedo, 0), kAmbient * unorm8(albedo, 1), kAmbient * unorm8(albedo, 2)};
        keep_result(g, i, rgb);
      }
      continue;
    }
    Pixel s[2];
    if (g != 0) {
      mat[0] = s_stash[kRowMaterial + 2][t];
      mat[1] = s_stash[kRowMaterial + 3][t];
      dep[0] = dep4.z;
      dep[1] = dep4.w;
      nlo[0] = s_stash[kRowNormal + 0][t];
      nhi[0] = s_stash[kRowNormal + 1][t];
      nlo[1] = s_stash[kRowNormal + 2][t];
      nhi[1] = s_stash[kRowNormal + 3][t];
      cx[0] = cx_next[0];
      cx[1] = cx_next[1];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) setup_pixel(s[i], p, cx[i], cy, dep[i], mat[i], nlo[i], nhi[i]);

    // shades the first `count` records of s_light (count even, > 0)
    auto shade_list = [&](uint32_t count) {
      const float4* rec = s_light;
      float4 pa = rec[0], ca = rec[1], pb = rec[2], cb = rec[3];
      for (uint32_t j = 0; j < count; j += 2u) {
        const Light A{pa.x, pa.y, pa.z, ca.x, ca.y, ca.z};
        const Light B{pb.x, pb.y, pb.z, cb.x, cb.y, cb.z};
        rec += 4;  // next pair (one pair beyond the list on the last trip: read, never used)
        pa = rec[0];
        ca = rec[1];
        pb = rec[2];
        cb = rec[3];
        shade4<kRadius>(s[0], s[1], A, B, kc);
      }
    };
    const float4 black_pos = make_float4(0.0f, 0.0f, 1e18f, 0.0f), black_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);

    if (!kTile) {
      for (uint32_t base = 0; base < L; base += kChunk) {
        const uint32_t n = min((uint32_t)kChunk, L - base);
        const uint32_t n2 = (n + 1u) & ~1u;
        __syncthreads();  // previous chunk fully consumed
        for (uint32_t i = t; i < n2; i += 256u) {
          float4 lp = black_pos, lc = black_col;
          if (i < n) {
            lp = lights[3u * (base + i)];       // pos.xyz, opening_angle (unused)
            lc = lights[3u * (base + i) + 1u];  // color.xyz, penumbra_angle (unused)
          }
          s_light[2u * i] = lp;
          s_light[2u * i + 1u] = lc;
        }
        __syncthreads();
        if (wave_live) shade_list(n2);
      }
    } else if (g != 0 && list_complete) {
      // the first pixel pair listed every reaching light without overflowing: the list is still in LDS
      if (wave_live && list_count > 0u) shade_list((list_count + 1u) & ~1u);
    } else {
      // Lights go through in spans of kTileSpan: pass 1 (test_span) leaves a ballot word and a list offset
      // per 64-light group; in pass 2 the hit lanes copy their records to the offsets, which keeps table
      // order (the float sums are order-sensitive).  A span with more hits than the list holds is shaded
      // in several rounds of whole groups.
      bool single_round = true;
      for (uint32_t span = 0; span < LL; span += kTileSpan) {
        const uint32_t span_lights = min((uint32_t)kTileSpan, LL - span);
        const uint32_t groups = (span_lights + 63u) >> 6;
        if (span != 0u || (g != 0 && LL > (uint32_t)kTileSpan)) test_span(span, std::integral_constant<uint32_t, 4u>());
        const uint32_t total = s_group_before[groups];
        uint32_t first = 0;  // first group of the round (block-uniform, like everything steering the barriers below)
        while (first < groups) {
          const uint32_t before = s_group_before[first];
          uint32_t end = groups;
          if (total - before > (uint32_t)kChunk) {
            end = first + 1u;  // one group always fits
            while (s_group_before[end + 1u] - before <= (uint32_t)kChunk) ++end;
            single_round = false;
          }
          const uint32_t count = s_group_before[end] - before;
          if (count > 0u) {
            for (uint32_t grp = first + wv; grp < end; grp += 4u) {
              const uint64_t hits = s_group_hits[grp];
              if ((hits >> lane) & 1ull) {
                const uint32_t pos = span + grp * 64u + lane;
                const uint32_t i = bin ? bin[pos] : pos;
                const uint32_t slot = s_group_before[grp] - before + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
                s_light[2u * slot] = lights[3u * i];
                s_light[2u * slot + 1u] = lights[3u * i + 1u];
              }
            }
            if (t == 0u && (count & 1u)) {
              s_light[2u * count] = black_pos;
              s_light[2u * count + 1u] = black_col;
            }
            __syncthreads();
            if (wave_live) shade_list((count + 1u) & ~1u);
            __syncthreads();  // everyone is done reading before the list is rebuilt (or reused)
          }
          first = end;
        }
        list_count = total;
      }
      list_complete = single_round && LL <= (uint32_t)kTileSpan;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float rgb[3];
      finish_pixel(s[i], s_stash[kRowAlbedo + g + i][t], s_stash[kRowMaterial + g + i][t], kRadius ? s[i].in_range > 0.0f : L > 0u, rgb);
      keep_result(g, i, rgb);
    }
  }
  if (!inside) return;

  const uint32_t valid = min(4u, p.width - x0);
  uint8_t* orow = p.out + (size_t)y * p.pitch_out;
  if (kOutMode == kOutRgba8) {
    uint32_t packed[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) packed[i] = s_stash[kRowAlbedo + i][t];
    uint32_t* o = reinterpret_cast<uint32_t*>(orow) + x0;
    if (valid == 4u) {
      *reinterpret_cast<uint4*>(o) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    } else {
#pragma unroll
      for (int i = 0; i < 3; ++i)
        if ((uint32_t)i < valid) o[i] = packed[i];
    }
  } else {
    float rgb[4][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int row = k == 0 ? kRowAlbedo : k == 1 ? kRowMaterial : kRowExtra;
        rgb[i][k] = __uint_as_float(s_stash[row + i][t]);
        rgb[2 + i][k] = last[i][k];
      }
    if (kOutMode == kOutRgba16f) {
      uint2* o = reinterpret_cast<uint2*>(orow) + x0;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if ((uint32_t)i < valid) {
          const __half2 a = __floats2half2_rn(rgb[i][0], rgb[i][1]);
          const __half2 b = __floats2half2_rn(rgb[i][2], 1.0f);
          o[i] = make_uint2(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b));
        }
      }
    } else {
      float4* o = reinterpret_cast<float4*>(orow) + x0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if ((uint32_t)i < valid) o[i] = make_float4(rgb[i][0], rgb[i][1], rgb[i][2], 1.0f);
    }
  }
}

// Device self-check of exact_div against the compiler's IEEE division over the operand ranges the
// resolve uses: counts mismatching bit patterns.
__global__ void exact_div_check_kernel(uint32_t n, uint32_t seed, uint32_t* mismatches) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (i + 1u) * 2654435761u ^ seed;
  auto next = [&]() {
    h ^= h << 13; h ^= h >> 17; h ^= h << 5;
    return (float)(h >> 8) * (1.0f / 16777216.0f);
  };
  const float u0 = next(), u1 = next(), u2 = next();
  // divisor: w of the unprojection (~1e-4 .. 16), an image extent (1 .. 65536), or any value in [2^-20, 2^20]
  float b, a;
  switch (i % 3u) {
    case 0: b = 1e-4f + 16.0f * u0 * u0; a = (u1 - 0.5f) * 4000.0f; break;
    case 1: b = floorf(1.0f + 65535.0f * u0); a = floorf(65536.0f * u1) + 0.5f; break;
    default: b = exp2f(40.0f * u0 - 20.0f) * (1.0f + u2); a = exp2f(40.0f * u1 - 20.0f) * (u2 < 0.5f ? -1.0f : 1.0f); break;
  }
  const float want = a / b;
  const float got = exact_div(a, exact_rcp(b));
  if (__float_as_uint(want) != __float_as_uint(got)) atomicAdd(mismatches, 1u);
}

// Coarse level of the tile light lists: one workgroup per kBinPixels x kBinPixels pixel bin lists, in table
// order, the lights whose sphere of influence reaches the bin's view pyramid (four side planes through the
// camera rays of the bin's corners; no depth bounds, so no G-buffer read).  A light is dropped only when it
// lies outside a side plane by more than radius + pad, where pad covers the binary32 error of the pixel
// positions it could light: those lie within the radius of the light, so their coordinates are bounded by
// |light coordinate| + radius (same 2e-3 relative margin as the per-tile bounds).
__global__ __launch_bounds__(256) void light_bins_kernel(const DeferredParams p, uint32_t* __restrict__ bin_list, uint32_t* __restrict__ bin_count) {
  __shared__ float s_face[16];
  __shared__ uint32_t s_cnt[4];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const uint32_t L = p.light_count;
  if (wv == 0u) {
    const uint32_t x0 = blockIdx.x * kBinPixels, y0 = blockIdx.y * kBinPixels;
    const float xa = (float)x0 / (float)p.width * 2.0f - 1.0f;
    const float xb = (float)min(x0 + kBinPixels, p.width) / (float)p.width * 2.0f - 1.0f;
    const float ya = (float)(y0 + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
    const float yb = (float)(min(y0 + kBinPixels, p.height) + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
    // lane 4*kind + c: corner c (xa ya, xb ya, xb yb, xa yb) at two depths of the same pixel ray
    const uint32_t c = lane & 3u, kind = (lane >> 2) & 1u;
    const float pcx = (c == 1u || c == 2u) ? xb : xa;
    const float pcy = (c >= 2u) ? yb : ya;
    const float pcz = kind == 0u ? 0.99f : 0.5f;
    const float* M = p.inv_vp;
    const float wx = ((M[0] * pcx + M[4] * pcy) + M[8] * pcz) + M[12];
    const float wy = ((M[1] * pcx + M[5] * pcy) + M[9] * pcz) + M[13];
    const float wz = ((M[2] * pcx + M[6] * pcy) + M[10] * pcz) + M[14];
    const float ww = ((M[3] * pcx + M[7] * pcy) + M[11] * pcz) + M[15];
    const float q[3] = {wx / ww, wy / ww, wz / ww};
    const bool ok = (ww > 1e-30f) && (fabsf(q[0]) < 1e30f) && (fabsf(q[1]) < 1e30f) && (fabsf(q[2]) < 1e30f);
    const bool sane = (__ballot(ok) & 0xFFull) == 0xFFull;
    float near1[3], near2[3], ray[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      near1[a] = __shfl(q[a], (int)((lane + 1u) & 3u));
      near2[a] = __shfl(q[a], (int)((lane + 2u) & 3u));
      ray[a] = __shfl(q[a], (int)(4u + c));
    }
    const float ux = ray[0] - q[0], uy = ray[1] - q[1], uz = ray[2] - q[2];
    const float vx = near1[0] - q[0], vy = near1[1] - q[1], vz = near1[2] - q[2];
    float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const float len2 = nx * nx + ny * ny + nz * nz;
    const float inv = len2 > 0.0f ? rsq_(len2) : 0.0f;  // degenerate face: zero normal, never culls
    nx *= inv; ny *= inv; nz *= inv;
    const float side = nx * (near2[0] - q[0]) + ny * (near2[1] - q[1]) + nz * (near2[2] - q[2]);
    const float flip = side > 0.0f ? -1.0f : 1.0f;
    const float fx = sane ? flip * nx : 0.0f, fy = sane ? flip * ny : 0.0f, fz = sane ? flip * nz : 0.0f;
    if (lane < 4u) {
      s_face[4u * lane + 0u] = fx;
      s_face[4u * lane + 1u] = fy;
      s_face[4u * lane + 2u] = fz;
      s_face[4u * lane + 3u] = -(fx * q[0] + fy * q[1] + fz * q[2]);
    }
  }
  __syncthreads();
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
  uint32_t* __restrict__ list = bin_list + (size_t)b * L;
  const float radius = sqrt_(p.radius2);
  uint32_t count = 0;  // block-uniform
  for (uint32_t base = 0; base < L; base += 256u) {
    const uint32_t i = base + t;
    bool hit = i < L;
    if (hit) {
      const float4 lp = lights[3u * i];
      const float scale = fmaxf(fmaxf(fabsf(lp.x), fabsf(lp.y)), fabsf(lp.z)) + radius;
      const float reach = radius + (2e-3f * scale + 1e-2f);
#pragma unroll
      for (int c = 0; c < 4; ++c) hit &= !(((s_face[4 * c] * lp.x + s_face[4 * c + 1] * lp.y) + s_face[4 * c + 2] * lp.z) + s_face[4 * c + 3] > reach);
    }
    const uint64_t hits = __ballot(hit);
    __syncthreads();  // every wave is past its previous read of s_cnt
    if (lane == 0u) s_cnt[wv] = (uint32_t)__popcll(hits);
    __syncthreads();
    const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    const uint32_t before = count + (wv > 0u ? c0 : 0u) + (wv > 1u ? c1 : 0u) + (wv > 2u ? c2 : 0u);
    if (hit) list[before + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u))] = i;
    count += (c0 + c1) + (c2 + c3);
  }
  if (t == 0u) bin_count[b] = count;
}

template <int kOutMode>
hipError_t launch_mode(const DeferredParams& p, hipStream_t stream) {
  const dim3 grid((p.width + 31u) / 32u, (p.height + 31u) / 32u, 1u);
  const bool radius = (p.flags & 1u) != 0u, no_tile = (p.flags & 2u) != 0u;
  if (!radius)
    hipLaunchKernelGGL((deferred_kernel<kOutMode, false, false>), grid, dim3(256), 0, stream, p);
  else if (no_tile)
    hipLaunchKernelGGL((deferred_kernel<kOutMode, true, false>), grid, dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((deferred_kernel<kOutMode, true, true>), grid, dim3(256), 0, stream, p);
  return hipGetLastError();
}

constexpr uint32_t kBinMinLights = 256;           // below this every tile tests the whole table: nothing to gain
constexpr uint64_t kBinMaxEntries = 64ull << 20;  // 256 MiB of indices

}  // namespace

hipError_t vkr_launch_exact_div_check(uint32_t n, uint32_t seed, uint32_t* mismatches, hipStream_t stream) {
  hipLaunchKernelGGL(exact_div_check_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, seed, mismatches);
  return hipGetLastError();
}

uint64_t vkr_deferred_bin_entries(const DeferredParams& p) {
  const bool tiled = (p.flags & 1u) != 0u && (p.flags & 2u) == 0u;
  if (!tiled || p.light_count < kBinMinLights) return 0;
  const uint64_t bins = (uint64_t)((p.width + kBinPixels - 1u) / kBinPixels) * ((p.height + kBinPixels - 1u) / kBinPixels);
  const uint64_t entries = bins * p.light_count;
  return entries <= kBinMaxEntries ? entries : 0;
}

hipError_t vkr_launch_deferred(const DeferredParams& params, uint32_t* bins, hipStream_t stream) {
  DeferredParams p = params;
  p.bin_list = nullptr;
  p.bin_count = nullptr;
  p.bins_x = 0;
  const uint64_t entries = vkr_deferred_bin_entries(p);
  if (bins && entries != 0u) {
    const dim3 bin_grid((p.width + kBinPixels - 1u) / kBinPixels, (p.height + kBinPixels - 1u) / kBinPixels, 1u);
    hipLaunchKernelGGL(light_bins_kernel, bin_grid, dim3(256), 0, stream, p, bins, bins + entries);
    if (hipError_t e = hipGetLastError()) return e;
    p.bin_list = bins;
    p.bin_count = bins + entries;
    p.bins_x = bin_grid.x;
  }
  switch (p.out_mode) {
    case kOutRgba8: return launch_mode<kOutRgba8>(p, stream);
    case kOutRgba16f: return launch_mode<kOutRgba16f>(p, stream);
    case kOutRgba32f: return launch_mode<kOutRgba32f>(p, stream);
    default: return hipErrorInvalidValue;
  }
}
