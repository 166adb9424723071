// Deferred lighting resolve for gfx950 (MI355X): the pass of
// /root/reference/shaders/deferred.comp:98-131, hand-written for wave64.
//
// This file is compiled with -ffp-contract=off: every fused multiply-add below is an explicit
// __builtin_fmaf, so the depth -> world-position step can reproduce the reference's binary32
// evaluation order bit for bit (it is ill-conditioned: w = a*z + b cancels as z -> 1), while the
// per-light loop is written in a cancellation-free form that needs 4 transcendental and 57 plain
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
// and F = Cs + (1 - Cs) (1 - v.h)^5 splits the specular sum into two colour-weighted sums, of which the Cs part
// joins the diffuse sum: diffuseColor = albedo (1 - m) and Cs = albedo m share the albedo, so with the per-pixel
// scalars A = (1 - m) PI ef and B = m Kspec
//   colour = ambient + albedo * sum_j c_j (A diffuse_j + B spec_j) + (Kspec - albedo B) * sum_j c_j spec_j (1 - v.h)^5
// - two sums of three channels per pixel, and A rides for nothing inside the diffuse weight's own multiply-adds.
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

namespace {

constexpr float kPi = 3.14159f;  // deferred.comp:3 (truncated on purpose)
constexpr float kAmbient = 0.05f;
constexpr float kAlphaFloor = 0.0064f;
constexpr float kEarlyOut = 1e-6f;
constexpr float kInvDistFloor = 1e7f;      // 1 / 1e-7, the floor of dist^2 (deferred.comp:122)
// The frame vectors T, B, N of a pixel are kept times sqrt(1e7): the light vector's components in the frame then
// come out times sqrt(1e7), the sum of their squares is dist^2 * 1e7 (the frame is orthonormal to rounding), its
// rsq is sqrt(1e-7) / dist, the cosines (component * rsq) are unscaled, and the distance floor of deferred.comp:122,
// min(1 / dist^2, 1e7) * 1e-7, is the clamp of rsq * rsq: one instruction.
constexpr float kFrameScale = 3162.27766f;
constexpr float kStepScale = 0x1p100f;     // (c - 1e-6) * 2^100 saturates the clamp for any c > 1e-6
constexpr float kRadiusStepScale = 0x1p80f;
constexpr int kChunk = 512;                // lights per LDS chunk: 2 float4 each = 16 KiB

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
  float tx, ty, tz;            // tangent frame, every vector times kFrameScale (see there)
  float bx, by, bz;
  float nx, ny, nz;            // shading normal = -normalize(gbuffer normal), times kFrameScale
  float vt, vb, vn;            // view direction in the (unscaled) frame
  // material
  float a2, rh, ebm1;          // alpha^2, roughness/2, energyBias-1
  float ka, kapv, kb;          // A = (1-metal) PI ef 1e7, A (1-vn)^5, B = metal Kspec 1e7 (see shade4 / finish_pixel)
  // sums over the lights, in table order, of colour_j times a scalar weight: s1 takes A*diffuse + B*spec (everything
  // the albedo multiplies), s2 takes spec*(1-v.h)^5; both weights carry the 1e-7 of the distance floor
  float s1[3], s2[3];
  float in_range;              // radius cutoff only: > 0 once any light lay inside the radius
};

struct Light {
  float px, py, pz, cx, cy, cz;
};

// Two lights applied to two pixels: the four (pixel, light) pairs k = 0..3 are (a,A) (b,A) (a,B) (b,B).
// Every weight carries a factor 1e-7 (the distance floor is applied as clamp(inv^2 * 1e-7), which
// equals min(1/d^2, 1e7) * 1e-7); finish_pixel() multiplies it back.
struct LoopConsts {
  float step_scale, step_bias, radius2_scaled;
};

// Wave-uniform state of the "every light of this trip is behind every surface of this wave" test.
// `budget` trips are still tested without a success; a success refills it.  A wave whose pixels face all ways
// (normals uncorrelated between neighbours) never succeeds and soon stops paying for the test; a wave over one
// surface succeeds for every light on the far side of it.  A list starts with a full budget if the wave's previous
// list ended with budget left, and with a token one otherwise (probe_budget_for).
struct Probe {
  uint32_t budget;
#ifdef VKR_X_COUNT
  uint32_t skipped = 0;
#endif
};
#ifdef VKR_X_TIMES  // measurement builds only (tools/wave_times.py): when each wave of the whole-table kernel started and ended, and where it ran
__device__ unsigned long long g_wave_times[4 * 65536];
#endif
#ifdef VKR_X_COUNT  // measurement builds only (tools/skip_stats.py): trips walked / skipped by the test / not walked at all (wave-level skip)
__device__ unsigned long long g_trip_counts[3];
#define VKR_COUNT_TRIPS(walked, skipped_by_test, not_walked)                                  \
  if ((threadIdx.x & 63u) == 0u) {                                                            \
    atomicAdd(&g_trip_counts[0], (unsigned long long)(walked));                               \
    atomicAdd(&g_trip_counts[1], (unsigned long long)(skipped_by_test));                      \
    atomicAdd(&g_trip_counts[2], (unsigned long long)(not_walked));                           \
  }
#else
#define VKR_COUNT_TRIPS(walked, skipped_by_test, not_walked)
#endif
#ifndef VKR_X_BUDGET
#define VKR_X_BUDGET 8
#endif
constexpr uint32_t kProbeBudget = VKR_X_BUDGET, kProbeBudgetLow = 2;
// Budget of a list of `trips` trips for a wave whose previous list left `left` (kProbeBudget before its first list).
__device__ __forceinline__ uint32_t probe_budget_for(uint32_t trips, uint32_t left) {
  return left == 0u ? kProbeBudgetLow : max(kProbeBudgetLow, min(kProbeBudget, trips / 8u));
}
// With the test (kProbe) the next trip's light records are requested from LDS inside the trip: the positions right
// after the test (it ends a basic block, and a request made ahead of it would pin its registers across it; the
// current positions are dead by then, so the new ones take their registers), the colours after the sums have
// taken the current ones.  Without the test the caller requests both ahead of the call, where the scheduler is free to
// place them.
// The four (pixel, light) pairs of a trip are two lights on two pixels (kLights = 2: pairs k = 0..3 are (a,L0) (b,L0)
// (a,L1) (b,L1)) or four lights on one pixel (kLights = 4: (a,L0) .. (a,L3); b is a).  Either way a pixel's sums take
// its lights in table order.
template <bool kRadius, bool kProbe, int kLights, class FetchPos, class FetchCol, bool kAhead = false>
__device__ __forceinline__ void shade4(Pixel& a, Pixel& b, const Light (&L)[kLights], const LoopConsts& kc, Probe& probe, FetchPos&& fetch_next_pos,
                                       FetchCol&& fetch_next_col) {
  static_assert(kLights == 2 || kLights == 4, "two lights on two pixels, or four on one");
  float dll[4], lt[4], lb[4], ln[4], inv[4], near[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
    const Light& l = L[kLights == 2 ? (k >> 1) : k];
    const float wx = l.px - s.px, wy = l.py - s.py, wz = l.pz - s.pz;
    lt[k] = fma_(wz, s.tz, fma_(wy, s.ty, wx * s.tx));
    lb[k] = fma_(wz, s.bz, fma_(wy, s.by, wx * s.bx));
    ln[k] = fma_(wz, s.nz, fma_(wy, s.ny, wx * s.nx));
    if (kRadius) {
      // the cutoff is decided on the light vector itself (the tile lists are conservative against exactly this sum)
      const float d2 = fma_(wz, wz, fma_(wy, wy, wx * wx));
      near[k] = clamp01_(fma_(d2, -kRadiusStepScale, kc.radius2_scaled));  // 1 iff dist^2 < radius^2
    }
    // dist^2 * 1e7 (kFrameScale), also with the cutoff: a radius beyond every light then gives the bytes of a call without one
    dll[k] = fma_(ln[k], ln[k], fma_(lb[k], lb[k], lt[k] * lt[k]));
  }
  rsq4(inv, dll);  // sqrt(1e-7) / dist
  float c[4], wu[4], wp[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c[k] = ln[k] * inv[k];
  bool skip = false;  // wave-uniform
  if (kProbe) {
    // deferred.comp:43,84 return 0 for l.n <= 1e-6.  When that holds for all four pairs of every lane, with every
    // cosine a number in [-2, -0] (as a signed integer the bit pattern of such a float is <= 0xC0000000; the
    // patterns of NaN, of -inf and of anything positive are greater), the step multiplier below is 0 and every other
    // factor is finite (positions, frames and 1/d are, or the cosine would not be a number of that size), so each of
    // the eight weights is +-0: the three remaining groups of transcendentals and the arithmetic between them are
    // skipped and the weights set to 0.  The sums still take colour * 0 (a colour that is not finite makes that NaN,
    // as it does on the long way round).
    const int m01 = max(__float_as_int(c[0]), __float_as_int(c[1])), m23 = max(__float_as_int(c[2]), __float_as_int(c[3]));
    const bool behind = max(m01, m23) <= (int)0xC0000000;
    skip = __ballot(!behind) == 0ull;
    probe.budget = skip ? kProbeBudget : probe.budget - 1u;
#ifdef VKR_X_COUNT
    probe.skipped += skip ? 1u : 0u;
#endif
    fetch_next_pos();
  }
  if (kAhead && !kProbe) fetch_next_pos();  // (the walk over per-pixel lists: the next trip's positions, once this trip's are used up)
  if (skip) {
    // as instructions of this branch: plain assignments are hoisted above the test, into every trip
#pragma unroll
    for (int k = 0; k < 4; ++k) asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(wu[k]), "=v"(wp[k]));
  } else {
    float hh[4], dn[4], att[4], sh[4], qa[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
      const float ht = fma_(lt[k], inv[k], s.vt);
      const float hb = fma_(lb[k], inv[k], s.vb);
      const float hn = c[k] + s.vn;
      const float ht2 = fma_(hb, hb, ht * ht);
      const float hn2 = hn * hn;
      hh[k] = ht2 + hn2;
      dn[k] = fma_(s.a2, hn2, ht2);
      att[k] = clamp01_(inv[k] * inv[k]);  // min(1/d^2, 1e7) * 1e-7
    }
    sqrt4(sh, hh);
    float akd[4], pvh[4], den[4], rden[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
      pvh[k] = pow5_(clamp01_(fma_(sh[k], -0.5f, 1.0f)));  // (1 - v.h)^5
      const float f90m1 = fma_(hh[k], s.rh, s.ebm1);
      const float pl = pow5_(clamp01_(1.0f - c[k]));
      const float ga = fma_(f90m1, s.kapv, s.ka);          // A * FS(1, f90, v.n)
      akd[k] = fma_(f90m1 * pl, ga, ga);                   // A * FS(v.n) * FS(l.n)   (deferred.comp:52-54)
      qa[k] = fma_(s.a2, c[k] * c[k], 1.0f);
    }
    sqrt4(q, qa);
#pragma unroll
    for (int k = 0; k < 4; ++k) den[k] = fma_(dn[k] * dn[k], c[k] + q[k], 1e-37f);  // +1e-37 keeps culled lanes finite
    rcp4(rden, den);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
      // early-out of deferred.comp:43,84 as a multiplier: (c - 1e-6) * 2^100 is exact inside the FMA; its clamp
      // is 1 for every c > 1e-6 and 0 for c <= 1e-6 or NaN
      float ci = (c[k] * clamp01_(fma_(c[k], kc.step_scale, kc.step_bias))) * att[k];
      if (kRadius) ci *= near[k];
      const float ks0 = (hh[k] * hh[k]) * rden[k];
      wu[k] = fma_(s.kb, ks0, akd[k]) * ci;  // what the albedo multiplies: A * diffuse + B * spec
      wp[k] = ks0 * (ci * pvh[k]);           // spec * (1 - v.h)^5
    }
  }
  if (kRadius) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
      s.in_range += near[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Pixel& s = (kLights == 2 && (k & 1)) ? b : a;
    const Light& l = L[kLights == 2 ? (k >> 1) : k];
    s.s1[0] = fma_(l.cx, wu[k], s.s1[0]);
    s.s1[1] = fma_(l.cy, wu[k], s.s1[1]);
    s.s1[2] = fma_(l.cz, wu[k], s.s1[2]);
    s.s2[0] = fma_(l.cx, wp[k], s.s2[0]);
    s.s2[1] = fma_(l.cy, wp[k], s.s2[1]);
    s.s2[2] = fma_(l.cz, wp[k], s.s2[2]);
  }
  if (kProbe) fetch_next_col();
}

// deferred.comp:99-100: clip-space coordinate of a pixel centre, true division included.
__device__ __forceinline__ float clip_coord(uint32_t i, const ExactRcp& by_extent) {
  return exact_div((float)i + 0.5f, by_extent) * 2.0f - 1.0f;
}

__device__ __forceinline__ float unorm8(uint32_t word, int byte) { return (float)((word >> (8 * byte)) & 0xFFu) * (1.0f / 255.0f); }

// Kspec * 1e7 = a2 / (PI (vn + sqrt(a2 vn^2 + 1))) * 1e7: the per-pixel factor of the specular term (file header).
__device__ __forceinline__ float spec_scale(const Pixel& s) {
  return (s.a2 * kInvDistFloor) * rcp_(kPi * (s.vn + sqrt_(fma_(s.a2, s.vn * s.vn, 1.0f))));
}

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

  // the light loop works with the frame times kFrameScale
  s.tx *= kFrameScale, s.ty *= kFrameScale, s.tz *= kFrameScale;
  s.bx *= kFrameScale, s.by *= kFrameScale, s.bz *= kFrameScale;
  s.nx *= kFrameScale, s.ny *= kFrameScale, s.nz *= kFrameScale;

  // :116 alpha = max(r*r, 0.0064)
  const float alpha = fmaxf(rough * rough, kAlphaFloor);
  s.a2 = alpha * alpha;
  s.rh = 0.5f * rough;
  s.ebm1 = fma_(0.5f, rough, -1.0f);
  // What multiplies a light's diffuse and specular weights per pixel, folded into the weights (the sums carry the
  // 1e-7 of the distance floor, so the 1e7 goes in here): with albedo_k the channel's albedo and m the metalness,
  //   colour_k = albedo_k * sum_j c_jk (A diffuse_j + B spec_j) + (Kspec - albedo_k B) * sum_j c_jk spec_j (1 - v.h)^5
  // (deferred.comp:114-115: diffuseColor = albedo (1 - m), specularColor = albedo m; :36-38: F = Cs + (1 - Cs)(1 - v.h)^5)
  const float metal = unorm8(material, 2);
  const float ef = fma_(const_inv151(), rough, 1.0f - rough);  // mix(1, 1/1.51, r)
  s.ka = (1.0f - metal) * ((kPi * kInvDistFloor) * ef);
  s.kapv = s.ka * pow5_(clamp01_(1.0f - s.vn));
  s.kb = metal * spec_scale(s);
#pragma unroll
  for (int k = 0; k < 3; ++k) s.s1[k] = s.s2[k] = 0.0f;
  s.in_range = 0.0f;
}

// Combines the two light sums into the pixel colour (deferred.comp:107,114-115,127-128; the formula is in the
// file header).  The albedo word is decoded here, after the light loop, so that nothing of it occupies registers
// during it.
// `lit_by_any`: at least one light took part for this pixel (any light at all, or with the radius
// cutoff one inside the radius).  A pixel whose normal or position is not a number (a cleared G-buffer
// texel has normal 0: normalize(0)) then comes out NaN, as every term of deferred.comp:120-129 does
// (rgba8 stores it as 0); with no light taking part it keeps its ambient colour.
__device__ __forceinline__ void finish_pixel(const Pixel& s, uint32_t albedo, uint32_t material, bool lit_by_any, float out[3]) {
  const bool front = s.vn > kEarlyOut;  // deferred.comp:43,84 (vDotN half of the early-out)
  const float unlit = (lit_by_any && !(s.vn == s.vn)) ? __builtin_nanf("") : 0.0f;
  const float kspec = spec_scale(s);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float alb = unorm8(albedo, k);
    const float lit = fma_(alb, s.s1[k], fma_(-alb, s.kb, kspec) * s.s2[k]);
    out[k] = fma_(kAmbient, alb, front ? lit : unlit);
  }
}

__device__ __forceinline__ uint32_t quantize_unorm8(float c) {
  // NaN -> 0, clamp, round-half-even (Vulkan float -> unorm conversion of the rgba8 store, :23,130)
  const float t = (c > 0.0f) ? fminf(c, 1.0f) : 0.0f;
  return (uint32_t)__builtin_rintf(t * 255.0f);
}

// One 8-bit texel, alpha 255: bytes R,G,B,A in memory for R8G8B8A8_UNORM, B,G,R,A for B8G8R8A8_UNORM (the
// reference's swapchain format, /root/reference/src/vkpresent.cpp:17: a store through a view of that format puts
// the colour's blue channel in byte 0).
template <int kOutMode>
__device__ __forceinline__ uint32_t pack_unorm8(const float (&rgb)[3]) {
  const uint32_t r = quantize_unorm8(rgb[0]), g = quantize_unorm8(rgb[1]), b = quantize_unorm8(rgb[2]);
  return kOutMode == kOutBgra8 ? (b | (g << 8) | (r << 16) | 0xFF000000u) : (r | (g << 8) | (b << 16) | 0xFF000000u);
}

// Shading cost class of a tile by the length of its list; the shading launch takes the classes from the top down.
__device__ __forceinline__ uint32_t list_class(uint32_t count) { return count < 8u ? 0u : count < 32u ? 1u : count < 128u ? 2u : 3u; }
static_assert(kListClasses == 4, "list_class() has four classes");

__device__ __forceinline__ uint4 load_stream_u4(const uint8_t* ptr) {
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(ptr));
  return make_uint4(v.x, v.y, v.z, v.w);
}

// Four consecutive pixels of an output row, `valid` of them inside the image.
__device__ __forceinline__ void store_rgba8x4(uint8_t* orow, uint32_t x0, uint32_t valid, const uint32_t (&packed)[4]) {
  uint32_t* o = reinterpret_cast<uint32_t*>(orow) + x0;
  if (valid == 4u) {
    *reinterpret_cast<uint4*>(o) = make_uint4(packed[0], packed[1], packed[2], packed[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i)
      if ((uint32_t)i < valid) o[i] = packed[i];
  }
}

// rgba16f texel of a binary32 colour, alpha 1: each component rounded to binary32 first and to half from there.
// The components pass through an empty asm statement because the compiler otherwise folds the colour's last
// multiply-add into the conversion (v_fma_mix*_f16: one rounding, straight to half) - closer to the exact value,
// but not the bits of "the binary32 result, converted", which is what the other kernels and the CPU evaluation
// produce; the two differ where the binary32 result lands on a half-way point (seen: 2 pixels of 230 400).
__device__ __forceinline__ uint2 half_texel(float r, float g, float b) {
  asm volatile("" : "+v"(r), "+v"(g), "+v"(b));
  const __half2 lo = __floats2half2_rn(r, g);
  const __half2 hi = __floats2half2_rn(b, 1.0f);
  return make_uint2(__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi));
}

template <int kOutMode>
__device__ __forceinline__ void store_float4x4(uint8_t* orow, uint32_t x0, uint32_t valid, const float (&rgb)[4][3]) {
  if (kOutMode == kOutRgba16f) {
    uint2* o = reinterpret_cast<uint2*>(orow) + x0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if ((uint32_t)i < valid) o[i] = half_texel(rgb[i][0], rgb[i][1], rgb[i][2]);
    }
  } else {
    float4* o = reinterpret_cast<float4*>(orow) + x0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if ((uint32_t)i < valid) o[i] = make_float4(rgb[i][0], rgb[i][1], rgb[i][2], 1.0f);
  }
}

// Work decomposition: a 256-thread workgroup shades a 32 x 32 pixel tile; lane t owns the four
// consecutive pixels (4*(t&7) .. +3, t>>3) of the tile, so every G-buffer plane is read with one
// 128-bit load per lane (two for the 8-byte normals) and the 8-bit result is written with one.
// The lane shades its four pixels as two pairs; each pair runs the light loop two lights at a time.
// The light records (pos.xyz, color.xyz as two float4) go through LDS in lists of up to kChunk,
// padded to an even count with a black light:
// One workgroup per tile of the launch grid walks the whole table, chunk after chunk (kRadius: with the cutoff
// applied per pixel; the production path of the cutoff is tile_lists_kernel + deferred_lists_kernel below).
// Ablations of the light loop at 4K x 256 lights (2.33 ms on the box that ran them): two trips' worth of
// lights per loop trip (half the loop's scalar instructions and branches per evaluation): 2.366 against 2.360 ms;
// no idle cycle after the fourth transcendental of a group (the s_setprio that follows already separates it from
// the next vector instruction): 2.331 against 2.331 ms.  The loop's non-vector instructions are not what it waits for.
// launch bound: 4 waves per SIMD (<= 128 VGPRs); measured 1-2 % faster than the 3 waves the allocator
// settles on unprompted.
// kWaves = waves per workgroup: the tile is 32 x (8 * kWaves) pixels and the LDS chunk 128 * kWaves lights.  Four
// waves (32 x 32) is the form in use, for launches of many rounds of workgroups; short launches (a screen band of a
// frame split over several GPUs, a 1080p frame) go to deferred_pixel_kernel below.
template <int kOutMode, bool kRadius, int kWaves>
__global__ __launch_bounds__(64 * kWaves, 4) void deferred_kernel(const DeferredParams p) {
  constexpr uint32_t kThreads = 64u * kWaves, kTileRows = 8u * kWaves, kLightChunk = 128u * kWaves;
  static_assert(kLightChunk <= (uint32_t)kChunk, "chunk beyond the staging area's size");
  __shared__ float4 s_light[2 * kLightChunk + 4];  // +1 pair: the prefetch of the last iteration reads past the list
  __shared__ uint32_t s_stash[12][kThreads];  // rows: see kRow* below
  constexpr bool kIs8Bit = kOutMode == kOutRgba8 || kOutMode == kOutBgra8;
  const uint32_t t = threadIdx.x;
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  const uint32_t L = p.light_count;
  LoopConsts kc;
  // loop constants pinned in VGPRs: as SGPR or literal operands they would halve the issue rate of their users
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.radius2_scaled) : "s"(p.radius2 * kRadiusStepScale));

#ifdef VKR_X_TIMES
  const unsigned long long t_start = __builtin_readcyclecounter();
  const unsigned long long rt_start = wall_clock64();
#endif
  const uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y;
  const uint32_t count = L;  // lights this tile walks
  // Where a lane's four pixels are.  Derived from the thread index wherever it is needed (through a copy the
  // compiler cannot see through) rather than kept: the light loop leaves no register for values that merely wait.
  struct Where {
    uint32_t x0, y, lx, ly;
    bool inside;
  };
  auto where = [&](uint32_t tq) {
    uint32_t ox = tile_x * 32u, oy = tile_y * kTileRows;  // uniform; opaque too, or their vector-register copies wait across the loop
    asm volatile("" : "+v"(tq), "+s"(ox), "+s"(oy));
    Where w;
    w.x0 = ox + (tq & 7u) * 4u;
    w.y = oy + (tq >> 3);
    w.inside = (w.x0 < p.width) && (w.y < p.height);
    // lanes outside the image shade the tile's first group (in bounds by construction) and store nothing
    w.lx = w.inside ? w.x0 : ox;
    w.ly = w.inside ? w.y : oy;
    return w;
  };
  const Where w0 = where(t);
  const uint32_t lx = w0.lx, ly = w0.ly;
  const bool inside = w0.inside;

  const uint4 alb4 = *reinterpret_cast<const uint4*>(p.albedo + (size_t)ly * p.pitch_albedo + (size_t)lx * 4u);
  const uint4 mat4 = *reinterpret_cast<const uint4*>(p.material + (size_t)ly * p.pitch_material + (size_t)lx * 4u);
  const uint4 nrmA = *reinterpret_cast<const uint4*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)lx * 8u);
  const uint4 nrmB = *reinterpret_cast<const uint4*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)lx * 8u + 16u);

  // Little more than what the light loop itself reads stays in registers across it.  The albedo and
  // material words of all four pixels (decoded after the loop), the normal words of the second pixel
  // pair and the first pair's results wait in LDS, one dword column per lane (conflict-free, private to
  // the lane: no barrier needed).  Left in VGPRs, the register allocator spills part of them to scratch
  // memory, which is HBM traffic of the order of the output itself.  The rows are kept few: past
  // 32 KiB of LDS per workgroup the kernel measurably slows down.
  enum { kRowAlbedo = 0, kRowMaterial = 4, kRowNormal = 8, kStashRows = 12 };
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

  const bool wave_live = __ballot(inside) != 0ull;  // wave-uniform

  uint32_t probe_left = kProbeBudget;  // wave-uniform: what the wave's previous list left of its test budget
  // One copy of the light loop serves both pixel pairs (no unrolling): half the code, same speed.
#pragma unroll 1
  for (int g = 0; g < 4; g += 2) {
    Pixel s[2];
    if (g != 0) {
      mat[0] = s_stash[kRowMaterial + 2][t];
      mat[1] = s_stash[kRowMaterial + 3][t];
      nlo[0] = s_stash[kRowNormal + 0][t];
      nhi[0] = s_stash[kRowNormal + 1][t];
      nlo[1] = s_stash[kRowNormal + 2][t];
      nhi[1] = s_stash[kRowNormal + 3][t];
    }
    {
      const Where w = where(t);
      // uniform values taken through copies the compiler cannot see through, so that they are fetched or derived
      // here and not kept in vector registers from before the light loop
      const uint8_t* depth = p.depth;
      uint32_t width = p.width, full_height = p.full_height;
      asm volatile("" : "+s"(depth), "+s"(width), "+s"(full_height));
      const float2 dep = *reinterpret_cast<const float2*>(depth + (size_t)w.ly * p.pitch_depth + (size_t)(w.lx + (uint32_t)g) * 4u);
      const ExactRcp by_width = exact_rcp((float)width), by_height = exact_rcp((float)full_height);
      const float cy = clip_coord(w.ly + p.row_offset, by_height);
      setup_pixel(s[0], p, clip_coord(w.lx + (uint32_t)g, by_width), cy, dep.x, mat[0], nlo[0], nhi[0]);
      setup_pixel(s[1], p, clip_coord(w.lx + (uint32_t)g + 1u, by_width), cy, dep.y, mat[1], nlo[1], nhi[1]);
    }

    // shades the first `n` records of s_light (n even, > 0)
    auto shade_list = [&](uint32_t n) {
      const float4* rec = s_light;
      float4 pa = rec[0], ca = rec[1], pb = rec[2], cb = rec[3];
      Probe probe{probe_budget_for(n / 2u, probe_left)};
      // Two copies of the loop: trips are tested for "every light behind every surface of the wave" while the
      // budget lasts, and the rest of the list runs through the loop without the test, which is the faster one per
      // trip (the test ends a basic block in the middle of the trip, and the register allocator pays for that
      // with copies).  The second loop starts from LDS again, so that nothing but the pixels ties the two together.
      uint32_t j = 0;
      for (; j < n && probe.budget != 0u; j += 2u) {
        const Light AB[2] = {{pa.x, pa.y, pa.z, ca.x, ca.y, ca.z}, {pb.x, pb.y, pb.z, cb.x, cb.y, cb.z}};
        rec += 4;  // next pair (one pair beyond the list on the last trip: read, never used)
        shade4<kRadius, true>(
            s[0], s[1], AB, kc, probe, [&] { pa = rec[0], pb = rec[2]; }, [&] { ca = rec[1], cb = rec[3]; });
      }
      if (j < n) {
        const float4* rec2 = s_light + 2u * j;
        float4 pa2 = rec2[0], ca2 = rec2[1], pb2 = rec2[2], cb2 = rec2[3];
        for (; j < n; j += 2u) {
          const Light AB[2] = {{pa2.x, pa2.y, pa2.z, ca2.x, ca2.y, ca2.z}, {pb2.x, pb2.y, pb2.z, cb2.x, cb2.y, cb2.z}};
          rec2 += 4;
          pa2 = rec2[0];
          ca2 = rec2[1];
          pb2 = rec2[2];
          cb2 = rec2[3];
          shade4<kRadius, false>(s[0], s[1], AB, kc, probe, [] {}, [] {});
        }
      }
      probe_left = probe.budget;
#ifdef VKR_X_COUNT
      VKR_COUNT_TRIPS(n / 2u, probe.skipped, 0u);
#endif
    };
    const float4 black_pos = make_float4(0.0f, 0.0f, 1e18f, 0.0f), black_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // deferred.comp:43,84: a pixel with v.n <= 1e-6 gets 0 from every light, and one whose v.n is not a number comes
    // out NaN whatever the loop computes (finish_pixel) - with the radius cutoff only if a light lay in range, which
    // the loop finds out.  A wave none of whose pixels the loop can change does not walk the table.
    const bool lane_lit = kRadius ? (!(s[0].vn <= kEarlyOut) || !(s[1].vn <= kEarlyOut)) : (s[0].vn > kEarlyOut || s[1].vn > kEarlyOut);
    const bool wave_lit = wave_live && __ballot(lane_lit) != 0ull;  // wave-uniform

    for (uint32_t base = 0; base < count; base += kLightChunk) {
      const uint32_t n = min(kLightChunk, count - base);
      const uint32_t n2 = (n + 1u) & ~1u;
      __syncthreads();  // previous chunk fully consumed
      for (uint32_t i = t; i < n2; i += kThreads) {
        float4 lp = black_pos, lc = black_col;
        if (i < n) {
          lp = lights[3u * (base + i)];       // pos.xyz, opening_angle (unused)
          lc = lights[3u * (base + i) + 1u];  // color.xyz, penumbra_angle (unused)
        }
        s_light[2u * i] = lp;
        s_light[2u * i + 1u] = lc;
      }
      __syncthreads();
      if (wave_lit) {
        shade_list(n2);
      } else if (wave_live) {
        VKR_COUNT_TRIPS(n2 / 2u, 0u, n2 / 2u);
      }
    }
    float rgb[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
      finish_pixel(s[i], s_stash[kRowAlbedo + g + i][t], s_stash[kRowMaterial + g + i][t], kRadius ? s[i].in_range > 0.0f : L > 0u, rgb[i]);
    if (kIs8Bit) {
      // 8-bit results wait in LDS (over the rows of the albedo words just consumed) until the second pair is
      // done, so that each lane writes its four pixels with one full-width store
#pragma unroll
      for (int i = 0; i < 2; ++i) s_stash[kRowAlbedo + g + i][t] = pack_unorm8<kOutMode>(rgb[i]);
    } else {
      // float texels are 8 or 16 bytes: a pair is a full-width store (or two) by itself
      const Where w = where(t);
      if (w.inside) {
        const uint32_t xg = w.x0 + (uint32_t)g;
        uint8_t* orow = p.out + (size_t)w.y * p.pitch_out;
        const bool first = xg < p.width, second = xg + 1u < p.width;
        if (kOutMode == kOutRgba16f) {
          uint2* o = reinterpret_cast<uint2*>(orow) + xg;
          const uint2 t0 = half_texel(rgb[0][0], rgb[0][1], rgb[0][2]), t1 = half_texel(rgb[1][0], rgb[1][1], rgb[1][2]);
          if (second)
            *reinterpret_cast<uint4*>(o) = make_uint4(t0.x, t0.y, t1.x, t1.y);
          else if (first)
            o[0] = t0;
        } else {
          float4* o = reinterpret_cast<float4*>(orow) + xg;
          if (first) o[0] = make_float4(rgb[0][0], rgb[0][1], rgb[0][2], 1.0f);
          if (second) o[1] = make_float4(rgb[1][0], rgb[1][1], rgb[1][2], 1.0f);
        }
      }
    }
  }
#ifdef VKR_X_TIMES
  if ((t & 63u) == 0u) {
    const uint32_t wave = (blockIdx.y * gridDim.x + blockIdx.x) * kWaves + (t >> 6);
    if (wave < 65536u) {
      uint32_t hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_wave_times[4u * wave + 0u] = rt_start;
      g_wave_times[4u * wave + 1u] = wall_clock64();
      g_wave_times[4u * wave + 2u] = ((unsigned long long)xcc << 32) | hw;
      g_wave_times[4u * wave + 3u] = __builtin_readcyclecounter() - t_start;
    }
  }
#endif
  if (kIs8Bit) {
    const Where w1 = where(t);
    if (w1.inside) {
      uint32_t packed[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) packed[i] = s_stash[kRowAlbedo + i][t];
      store_rgba8x4(p.out + (size_t)w1.y * p.pitch_out, w1.x0, min(4u, p.width - w1.x0), packed);
    }
  }
}

// ------------------------------------------------------------------ whole table, one pixel per lane
// For launches of a few rounds of workgroups (a screen band of a frame split over several GPUs, a 1080p frame).
// Three waves of this code fill a SIMD's issue slots and the arbiter serves equal priorities oldest first, so the
// fourth wave of a SIMD hardly moves until one of the others ends; the SIMD's last wave then runs alone, at half the
// rate a full SIMD sustains.  While workgroups keep arriving that costs nothing, but at the end of a launch it adds
// about a ninth of a wave's lifetime (measured, 4K x 256 lights: 36 us per launch with 4 pixels per lane, 19 us with
// waves half as long, 12 us with a quarter).  So the short launches use the shortest waves: a single-wave workgroup
// shades 64 consecutive pixels of a row, one per lane, and walks the table once, four lights per trip - the same four
// (pixel, light) evaluations per trip, and per pixel the same sums in the same order, as the kernels above: the
// bytes are the same.
#ifndef VKR_X_PIXEL_CHUNK
#define VKR_X_PIXEL_CHUNK 128
#endif
constexpr uint32_t kPixelChunk = VKR_X_PIXEL_CHUNK;  // lights per LDS chunk

template <int kOutMode, bool kRadius>
__global__ __launch_bounds__(64, 4) void deferred_pixel_kernel(const DeferredParams p) {
  __shared__ float4 s_light[2 * kPixelChunk + 8];  // + one trip: the prefetch of the last trip reads past the list
  constexpr bool kIs8Bit = kOutMode == kOutRgba8 || kOutMode == kOutBgra8;
  const uint32_t t = threadIdx.x;
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  const uint32_t L = p.light_count;
  LoopConsts kc;
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.radius2_scaled) : "s"(p.radius2 * kRadiusStepScale));

  const uint32_t x = blockIdx.x * 64u + t, y = blockIdx.y;
  const bool inside = x < p.width;
  // lanes beyond the row's end shade the workgroup's first pixel (in bounds by construction) and store nothing
  const uint32_t lx = inside ? x : blockIdx.x * 64u;
  const uint32_t alb = *reinterpret_cast<const uint32_t*>(p.albedo + (size_t)y * p.pitch_albedo + (size_t)lx * 4u);
  const uint32_t mat = *reinterpret_cast<const uint32_t*>(p.material + (size_t)y * p.pitch_material + (size_t)lx * 4u);
  const float dep = *reinterpret_cast<const float*>(p.depth + (size_t)y * p.pitch_depth + (size_t)lx * 4u);
  const uint2 nrm = *reinterpret_cast<const uint2*>(p.normal + (size_t)y * p.pitch_normal + (size_t)lx * 8u);
  Pixel s;
  {
    const ExactRcp by_width = exact_rcp((float)p.width), by_height = exact_rcp((float)p.full_height);
    setup_pixel(s, p, clip_coord(lx, by_width), clip_coord(y + p.row_offset, by_height), dep, mat, nrm.x, nrm.y);
  }
  // deferred.comp:43,84: see deferred_kernel - a wave none of whose pixels the loop can change does not walk the table
  const bool lane_lit = kRadius ? !(s.vn <= kEarlyOut) : (s.vn > kEarlyOut);
  const bool wave_lit = __ballot(lane_lit) != 0ull;  // wave-uniform
  const float4 black_pos = make_float4(0.0f, 0.0f, 1e18f, 0.0f), black_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  uint32_t probe_left = kProbeBudget;
  for (uint32_t base = 0; base < L; base += kPixelChunk) {
    const uint32_t n = min(kPixelChunk, L - base);
    const uint32_t n4 = (n + 3u) & ~3u;  // padded to whole trips with black lights
    // one wave: its LDS accesses execute in program order, so only the compiler has to be kept from reordering them
    for (uint32_t i = t; i < n4; i += 64u) {
      float4 lp = black_pos, lc = black_col;
      if (i < n) {
        lp = lights[3u * (base + i)];       // pos.xyz, opening_angle (unused)
        lc = lights[3u * (base + i) + 1u];  // color.xyz, penumbra_angle (unused)
      }
      s_light[2u * i] = lp;
      s_light[2u * i + 1u] = lc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (wave_lit) {
      Probe probe{probe_budget_for(n4 / 4u, probe_left)};
      const float4* rec = s_light;
      float4 q0 = rec[0], c0 = rec[1], q1 = rec[2], c1 = rec[3], q2 = rec[4], c2 = rec[5], q3 = rec[6], c3 = rec[7];
      uint32_t j = 0;
      for (; j < n4 && probe.budget != 0u; j += 4u) {  // see deferred_kernel: the loop with the test, then the one without
        const Light four[4] = {{q0.x, q0.y, q0.z, c0.x, c0.y, c0.z}, {q1.x, q1.y, q1.z, c1.x, c1.y, c1.z},
                               {q2.x, q2.y, q2.z, c2.x, c2.y, c2.z}, {q3.x, q3.y, q3.z, c3.x, c3.y, c3.z}};
        rec += 8;  // next trip (one trip beyond the list on the last one: read, never used)
        shade4<kRadius, true>(
            s, s, four, kc, probe, [&] { q0 = rec[0], q1 = rec[2], q2 = rec[4], q3 = rec[6]; }, [&] { c0 = rec[1], c1 = rec[3], c2 = rec[5], c3 = rec[7]; });
      }
      if (j < n4) {
        // the record index lives in a vector register the compiler cannot see through: one address and eight
        // immediate offsets per trip, where it would otherwise keep eight scalar addresses and copy each to a
        // vector register for its read
        uint32_t at = 2u * j;
        asm volatile("" : "+v"(at));
        const float4* rec2 = s_light + at;
        float4 r0 = rec2[0], d0 = rec2[1], r1 = rec2[2], d1 = rec2[3], r2 = rec2[4], d2 = rec2[5], r3 = rec2[6], d3 = rec2[7];
        for (; j < n4; j += 4u) {
          const Light four[4] = {{r0.x, r0.y, r0.z, d0.x, d0.y, d0.z}, {r1.x, r1.y, r1.z, d1.x, d1.y, d1.z},
                                 {r2.x, r2.y, r2.z, d2.x, d2.y, d2.z}, {r3.x, r3.y, r3.z, d3.x, d3.y, d3.z}};
          at += 8u;
          asm volatile("" : "+v"(at));
          rec2 = s_light + at;
          r0 = rec2[0], d0 = rec2[1], r1 = rec2[2], d1 = rec2[3], r2 = rec2[4], d2 = rec2[5], r3 = rec2[6], d3 = rec2[7];
          shade4<kRadius, false>(s, s, four, kc, probe, [] {}, [] {});
        }
      }
      probe_left = probe.budget;
#ifdef VKR_X_COUNT
      VKR_COUNT_TRIPS(n4 / 4u, probe.skipped, 0u);
#endif
    } else {
      VKR_COUNT_TRIPS(n4 / 4u, 0u, n4 / 4u);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  float rgb[3];
  finish_pixel(s, alb, mat, kRadius ? s.in_range > 0.0f : L > 0u, rgb);
  if (inside) {
    uint8_t* orow = p.out + (size_t)y * p.pitch_out;
    if (kIs8Bit) {
      reinterpret_cast<uint32_t*>(orow)[x] = pack_unorm8<kOutMode>(rgb);
    } else if (kOutMode == kOutRgba16f) {
      reinterpret_cast<uint2*>(orow)[x] = half_texel(rgb[0], rgb[1], rgb[2]);
    } else {
      reinterpret_cast<float4*>(orow)[x] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
    }
  }
}

// ------------------------------------------------------------------ whole table, each pixel walks only its own lights
// deferred.comp:43,84 return 0 for a light with l.n <= 1e-6: in a frame whose normals face all ways (the headline's
// synthetic G-buffer) two thirds of the (pixel, light) evaluations end there, but hardly ever for all 64 pixels of a
// wave at once, so the wave-uniform tests of the kernels above find nothing to skip.  This kernel gives every pixel
// its own list: a bit per light, set unless a cheap and provably one-sided test says the light is behind the
// pixel's surface; each lane then walks the set bits of its own mask, four lights per trip, fetching its lights'
// records from LDS by its own indices.  A skipped light is one whose weights the long way round would be exactly
// +-0 and whose colour is finite, so the sums - taken in table order either way - come out bit for bit the same.
//
// The test: with n' the scaled shading normal, l.n' - p.n' + margin < 0, three multiply-adds per (pixel, light)
// on the light's position (no light vector, no rsq), its sign shifted into the mask by one v_alignbit.  The
// margin, 2^-20 * kFrameScale * (|p|_1 + largest |l|_1 of the table), exceeds the rounding of this sum and of the
// long way's (l - p).n' together (each below 2^-22 * kFrameScale * (|l|_1 + |p|_1)), so a set sign bit means the long
// way's l.n' is negative, not tiny, and its cosine a number in [-1, -0].  Lights that are not finite, or beyond 1e15
// (the long way's sums overflow there), are walked by every pixel (force), pixels beyond 1e15 walk every light, and
// pixels no light can change (v.n <= 1e-6 or not a number) walk none.  The positions come through scalar loads (the
// index is uniform): as LDS reads they would be 12 bytes per lane and light for four instructions' worth of work, three
// times what a CU's LDS delivers.
//
// Lanes that walk lists of different lengths wait for the longest, and the lengths differ widely (a pixel facing
// the bulk of the lights passes 70 % of them, one facing away 10 %), so the pixels of a tile are sorted by list
// length (counting sort over 128 buckets in LDS) and dealt out in that order: a wave shades batches of 64 pixels of
// like lengths, taken so that the four waves of the workgroup end up with like totals (batches w, 7 - w, 8 + w,
// 15 - w of the order).  Unsorted the kernel is slower than the loops that evaluate everything (2.76 against 2.25 ms).
// One workgroup per tile of 32 x 16 or 32 x 32 pixels (four or eight waves, every lane prepares two pixels and shades
// two); a table of more than kWalkLights lights is taken that many at a time (kChunked).
// Measurement builds (tools/build_variants.sh, tools/ab_deferred.py): -DVKR_X_WALK_NOMASK builds empty lists,
// -DVKR_X_WALK_NOWALK builds the lists and walks none, -DVKR_X_WALK_NOSORT deals the pixels out unsorted,
// -DVKR_X_WALK_SMALL_TILES=1 uses the small tile for every launch, -DVKR_X_WALK_MIN=n moves the light count from which
// the kernel is used.
constexpr uint32_t kWalkLights = 256;
constexpr uint32_t kWalkWords = kWalkLights / 32u;
constexpr uint32_t kWalkStride = 7;   // dwords per staged record (px py pz cx cy cz, one unused): 7 is coprime with the 32 banks
constexpr float kWalkMargin = 0x1p-20f * kFrameScale;
constexpr float kWalkFar = 1e15f;
constexpr uint32_t kWalkHome = 2;  // consecutive pixels of a row that a lane prepares; a wave shades as many batches

// The first 16 bytes of eight consecutive 48-byte light records at a wave-uniform address, through the scalar cache
// into scalar registers: one address, eight immediate offsets, one wait.  (Written as indexed loads the compiler
// splits each record into three one-dword loads with an address computation each and waits after every second one.)
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void scalar_load_8_records(const float4* base, v4f (&q)[8]) {
  asm volatile(
      "s_load_dwordx4 %0, %8, 0x0\n\t"
      "s_load_dwordx4 %1, %8, 0x30\n\t"
      "s_load_dwordx4 %2, %8, 0x60\n\t"
      "s_load_dwordx4 %3, %8, 0x90\n\t"
      "s_load_dwordx4 %4, %8, 0xc0\n\t"
      "s_load_dwordx4 %5, %8, 0xf0\n\t"
      "s_load_dwordx4 %6, %8, 0x120\n\t"
      "s_load_dwordx4 %7, %8, 0x150\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(q[0]), "=&s"(q[1]), "=&s"(q[2]), "=&s"(q[3]), "=&s"(q[4]), "=&s"(q[5]), "=&s"(q[6]), "=&s"(q[7])
      : "s"(base)
      : "memory");
}

// The sign of `v` shifted into `mask` from below: light j of a block of 32 ends up in bit 31 - j.
__device__ __forceinline__ uint32_t push_sign(uint32_t mask, float v) { return __builtin_amdgcn_alignbit(mask, __float_as_uint(v), 31u); }

template <uint32_t kWalkPixels, bool kChunked>
struct WalkLds {
  float rec[(kWalkLights + 1u) * kWalkStride];  // first, so that a record's dwords are immediate offsets from its address; + the black light, which lanes whose list has ended walk
  uint32_t mask[kWalkWords][kWalkPixels];       // word w of pixel id's list
  uint32_t force[kWalkWords];
  uint32_t hist[128];
  uint32_t lmax;
  uint32_t out[kWalkPixels];                    // 8-bit results by pixel (row << 5 | column within the tile)
  uint16_t order[kWalkPixels];
  uint16_t length[kWalkPixels];                 // lights on each pixel's list
  float sums[kChunked ? 6 : 1][kChunked ? kWalkPixels : 1];  // tables of several chunks: every pixel's two sums between chunks
};

// kWalkRows = rows of a tile = 16 (four waves) or 32 (eight waves; more pixels to sort, so batches of more even
// lengths: 24.2 instead of 26.1 trips per batch on the headline frame, where lists of the mean length would take 21.8;
// as many waves per CU either way)
// kChunked: the table is longer than kWalkLights and is taken in chunks (below).
template <int kOutMode, uint32_t kWalkRows, bool kChunked>
__global__ __launch_bounds__(16 * kWalkRows, kChunked ? 4 : 1024 / (16 * kWalkRows)) void deferred_walk_kernel(const DeferredParams p) {
  constexpr uint32_t kWalkPixels = 32u * kWalkRows, kThreads = 16u * kWalkRows, kWaves = kThreads / 64u;
  __shared__ WalkLds<kWalkPixels, kChunked> lds;
  constexpr bool kIs8Bit = kOutMode == kOutRgba8 || kOutMode == kOutBgra8;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t tile_x = blockIdx.x, tile_y = blockIdx.y;
#ifdef VKR_X_TIMES  // measurement builds only (tools/walk_times.py): the wall clock when a wave starts, when it has its lists and order, and when it ends
  const unsigned long long rt_start = wall_clock64();
  unsigned long long rt_lists = 0ull;
#endif
  // A table of more than kWalkLights lights is taken that many at a time (kChunked).  The order in which the pixels
  // are dealt out is that of the first chunk's lists (a pixel that faces the bulk of one chunk's lights faces the bulk
  // of the next one's), so a pixel stays with its lane, and its two sums wait in LDS from one chunk to the next.
#pragma unroll 1
  for (uint32_t chunk0 = 0; chunk0 < (kChunked ? p.light_count : 1u); chunk0 += kWalkLights) {
  const uint32_t L = min(kWalkLights, p.light_count - chunk0);  // 1 .. kWalkLights
  const bool first_chunk = !kChunked || chunk0 == 0u, last_chunk = !kChunked || chunk0 + kWalkLights >= p.light_count;
  const uint32_t words = (L + 31u) / 32u;
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights) + 3u * chunk0;

  // (several chunks: derived per chunk, through copies the compiler cannot see through, rather than kept in registers
  // across the chunks)
  uint32_t width = p.width, full_height = p.full_height;
  if (kChunked) asm volatile("" : "+s"(width), "+s"(full_height));
  const ExactRcp by_width = exact_rcp((float)width), by_height = exact_rcp((float)full_height);

  // the texels of the pixels this lane prepares are requested ahead of the table's records: the two round trips overlap
  // (one chunk only: with several the registers are needed across the walks)
  float home_dep[kWalkHome];
  uint32_t home_mat[kWalkHome];
  uint2 home_nrm[kWalkHome];
  auto fetch_home = [&](uint32_t tq) {
    const uint32_t xl = (tq & 15u) * kWalkHome, yl = tq >> 4;
    const uint32_t x0 = tile_x * 32u + xl, y = tile_y * kWalkRows + yl;
    const bool row_in = y < p.height && x0 < p.width;
    const uint32_t lx = row_in ? x0 : tile_x * 32u, ly = row_in ? y : tile_y * kWalkRows;  // in bounds by construction
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) {
      const uint32_t xi = min(lx + i, p.width - 1u);
      home_dep[i] = *reinterpret_cast<const float*>(p.depth + (size_t)ly * p.pitch_depth + (size_t)xi * 4u);
      home_mat[i] = *reinterpret_cast<const uint32_t*>(p.material + (size_t)ly * p.pitch_material + (size_t)xi * 4u);
      home_nrm[i] = *reinterpret_cast<const uint2*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)xi * 8u);
    }
  };
  if (!kChunked) fetch_home(t);

  // ---- the chunk into LDS; which lights every pixel must walk; the largest |l|_1 of the others
  if (!first_chunk) __syncthreads();  // the walks over the previous chunk's records and lists have ended
  if (t == 0u) lds.lmax = 0u;
  if (first_chunk && t < 128u) lds.hist[t] = 0u;
  __syncthreads();
  if (t < kWalkLights) {  // the first four waves, one light per lane
    float4 lp = make_float4(0.0f, 0.0f, 1e18f, 0.0f), lc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    bool forced = false;
    if (t < L) {
      lp = lights[3u * t];
      lc = lights[3u * t + 1u];
      const float l1 = (fabsf(lp.x) + fabsf(lp.y)) + fabsf(lp.z), c1 = (fabsf(lc.x) + fabsf(lc.y)) + fabsf(lc.z);
      forced = !(l1 < kWalkFar) || !(c1 < INFINITY);  // also true for NaN
      if (!forced) atomicMax(&lds.lmax, __float_as_uint(l1));
    }
    float* r = lds.rec + t * kWalkStride;
    r[0] = lp.x, r[1] = lp.y, r[2] = lp.z, r[3] = lc.x, r[4] = lc.y, r[5] = lc.z;
    if (t == 0u) {
      float* b = lds.rec + kWalkLights * kWalkStride;
      b[0] = 0.0f, b[1] = 0.0f, b[2] = 1e18f, b[3] = 0.0f, b[4] = 0.0f, b[5] = 0.0f;
    }
    const uint64_t f = __ballot(forced);
    if (lane == 0u) {
      lds.force[2u * wave] = __brev((uint32_t)f);
      lds.force[2u * wave + 1u] = __brev((uint32_t)(f >> 32));
    }
  }
  __syncthreads();
  const float lmax = __uint_as_float(lds.lmax);
  // bits of word w that stand for lights of the table
  auto valid_bits = [&](uint32_t w) -> uint32_t { return (w + 1u) * 32u <= L ? 0xFFFFFFFFu : (w * 32u < L ? 0xFFFFFFFFu << (32u - (L & 31u)) : 0u); };

  // ---- every pixel's list; lane t prepares the kWalkHome consecutive pixels of its row
  {
    uint32_t tq = t;
    if (kChunked) asm volatile("" : "+v"(tq));  // what follows is the same for every chunk: derived again, not kept across the walks
    const uint32_t xl = (tq & 15u) * kWalkHome, yl = tq >> 4;
    const uint32_t x0 = tile_x * 32u + xl, y = tile_y * kWalkRows + yl;
    const bool row_in = y < p.height && x0 < p.width;
    const uint32_t lx = row_in ? x0 : tile_x * 32u, ly = row_in ? y : tile_y * kWalkRows;  // in bounds by construction
    const float cy = clip_coord(ly + p.row_offset, by_height);
    // a pixel's side of the test: the scaled normal and cst = margin - p.n'; kind: 0 walks none, 1 tests, 2 walks all
    float nx[kWalkHome], ny[kWalkHome], nz[kWalkHome], cst[kWalkHome];
    uint32_t kind[kWalkHome], count[kWalkHome];
    if (kChunked) fetch_home(tq);
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) {
      const uint32_t xi = min(lx + i, p.width - 1u);
      Pixel s;
      setup_pixel(s, p, clip_coord(xi, by_width), cy, home_dep[i], home_mat[i], home_nrm[i].x, home_nrm[i].y);
      nx[i] = s.nx, ny[i] = s.ny, nz[i] = s.nz;
      const float p1 = (fabsf(s.px) + fabsf(s.py)) + fabsf(s.pz);
      const float pn = fma_(s.pz, s.nz, fma_(s.py, s.ny, s.px * s.nx));
      cst[i] = fma_(kWalkMargin, p1 + lmax, 1e-15f) - pn;
      const bool in_image = row_in && x0 + i < p.width;
      kind[i] = !(in_image && s.vn > kEarlyOut) ? 0u : (p1 < kWalkFar ? 1u : 2u);
      count[i] = 0u;
    }
    // a wave none of whose pixels a light can change (sky, back faces) writes its empty lists and skips the tests
    bool wave_live = false;
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) wave_live = wave_live || __ballot(kind[i] != 0u) != 0ull;
#ifdef VKR_X_WALK_NOMASK
    wave_live = false;
#endif
    if (!wave_live) {
      for (uint32_t w = 0; w < words; ++w)
        for (uint32_t i = 0; i < kWalkHome; ++i) lds.mask[w][yl * 32u + xl + i] = 0u;
    }
    for (uint32_t w = 0; w < (wave_live ? words : 0u); ++w) {
      uint32_t raw[kWalkHome];
#pragma unroll
      for (uint32_t i = 0; i < kWalkHome; ++i) raw[i] = 0u;
      // always 32 lights (beyond the table's end its last light again: those bits are masked off below), eight
      // positions requested at a time so that their round trips overlap; a whole block of eight is one address and
      // eight immediate offsets
      for (uint32_t j0 = 0; j0 < 32u; j0 += 8u) {
        float qx[8], qy[8], qz[8];
        const uint32_t first = w * 32u + j0;
        if (first + 8u <= L) {
          v4f q[8];
          scalar_load_8_records(lights + 3u * first, q);
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) qx[j] = q[j].x, qy[j] = q[j].y, qz[j] = q[j].z;
        } else {
#pragma unroll
          for (uint32_t j = 0; j < 8u; ++j) {
            const float4 q = lights[3u * min(first + j, L - 1u)];
            qx[j] = q.x, qy[j] = q.y, qz[j] = q.z;
          }
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
#pragma unroll
          for (uint32_t i = 0; i < kWalkHome; ++i) raw[i] = push_sign(raw[i], fma_(qz[j], nz[i], fma_(qy[j], ny[i], fma_(qx[j], nx[i], cst[i]))));
        }
      }
      const uint32_t valid = valid_bits(w), force = lds.force[w];
#pragma unroll
      for (uint32_t i = 0; i < kWalkHome; ++i) {
        const uint32_t tested = (~raw[i] | force) & valid;
        const uint32_t word = kind[i] == 0u ? 0u : (kind[i] == 1u ? tested : valid);
        lds.mask[w][yl * 32u + xl + i] = word;
        count[i] += __popc(word);
      }
    }
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) lds.length[yl * 32u + xl + i] = (uint16_t)count[i];
    // counting sort by length / 2 (128 buckets; equal lengths stay together, their order among themselves is arbitrary
    // and changes nothing: a pixel's result does not depend on the lane that shades it)
    if (first_chunk) {
    uint32_t bucket[kWalkHome], within[kWalkHome];
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) {
      bucket[i] = min(count[i], 255u) >> 1;
      within[i] = atomicAdd(&lds.hist[bucket[i]], 1u);
    }
    __syncthreads();
    if (wave == 0u) {  // exclusive prefix over the 128 buckets: two per lane
      const uint32_t a = lds.hist[2u * lane], b = lds.hist[2u * lane + 1u];
      uint32_t incl = a + b;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += up;
      }
      lds.hist[2u * lane] = incl - (a + b);
      lds.hist[2u * lane + 1u] = incl - b;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kWalkHome; ++i) {
#ifdef VKR_X_WALK_NOSORT
      lds.order[t * kWalkHome + i] = (uint16_t)(yl * 32u + xl + i);
#else
      lds.order[lds.hist[bucket[i]] + within[i]] = (uint16_t)(yl * 32u + xl + i);
#endif
    }
    }
    __syncthreads();
  }

#ifdef VKR_X_TIMES
  rt_lists = wall_clock64();
#endif
  LoopConsts kc;
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  kc.radius2_scaled = 0.0f;
  Probe probe{0u};

  // ---- two batches of 64 pixels per wave: one from each end of the order.  Both batches' texels are requested here:
  // the second batch's arrive while the first is shaded.
  struct Texels {
    uint32_t id, alb, mat;
    float dep;
    uint2 nrm;
  };
  auto fetch = [&](uint32_t k) {
    Texels q;
    q.id = lds.order[64u * (k == 0u ? wave : 2u * kWaves - 1u - wave) + lane];
    const uint32_t x = tile_x * 32u + (q.id & 31u), y = tile_y * kWalkRows + (q.id >> 5);
    const bool inside = x < p.width && y < p.height;
    const uint32_t lx = inside ? x : tile_x * 32u, ly = inside ? y : tile_y * kWalkRows;
    q.alb = *reinterpret_cast<const uint32_t*>(p.albedo + (size_t)ly * p.pitch_albedo + (size_t)lx * 4u);
    q.mat = *reinterpret_cast<const uint32_t*>(p.material + (size_t)ly * p.pitch_material + (size_t)lx * 4u);
    q.dep = *reinterpret_cast<const float*>(p.depth + (size_t)ly * p.pitch_depth + (size_t)lx * 4u);
    q.nrm = *reinterpret_cast<const uint2*>(p.normal + (size_t)ly * p.pitch_normal + (size_t)lx * 8u);
    return q;
  };
  Texels ahead = fetch(0u);
#pragma unroll 1
  for (uint32_t k = 0; k < kWalkHome; ++k) {
    const Texels q = ahead;
    if (k + 1u < kWalkHome) ahead = fetch(k + 1u);
    const uint32_t id = q.id, alb = q.alb, mat = q.mat;
    const float dep = q.dep;
    const uint2 nrm = q.nrm;
    const uint32_t x = tile_x * 32u + (id & 31u), y = tile_y * kWalkRows + (id >> 5);
    const bool inside = x < p.width && y < p.height;
    const uint32_t lx = inside ? x : tile_x * 32u, ly = inside ? y : tile_y * kWalkRows;
    Pixel s;
    setup_pixel(s, p, clip_coord(lx, by_width), clip_coord(ly + p.row_offset, by_height), dep, mat, nrm.x, nrm.y);
    if (kChunked && !first_chunk) {
#pragma unroll
      for (int i = 0; i < 3; ++i) s.s1[i] = lds.sums[i][id], s.s2[i] = lds.sums[3 + i][id];
    }

    const uint32_t* my_words = &lds.mask[0][id];  // word w at my_words[w * kWalkPixels]
    // `left` = lights of the list not yet taken (may go below zero: the last trip's spare places take the black light);
    // cur = the word being walked (bits of lights not yet taken), wi = index of the next word
    int32_t left = (int32_t)lds.length[id];
    uint32_t cur = 0u, wi = 0u;
#ifdef VKR_X_WALK_NOWALK
    left = 0;
#endif
    // Taking the next four lights off the list: their records' byte offsets, and the requests for their positions.
    uint32_t at[4];
    float3 pos4[4];
    auto take_four = [&]() {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool has = left > e;
        if (has && cur == 0u) {
          do {  // a block of 32 lights none of which is on the list has an empty word
            cur = my_words[wi * kWalkPixels];
            ++wi;
          } while (cur == 0u);
        }
        const uint32_t pos = (uint32_t)__clz((int)cur) & 31u;
        at[e] = has ? ((wi - 1u) * 32u + pos) * (kWalkStride * 4u) : kWalkLights * kWalkStride * 4u;
        cur = has ? (cur & ~(0x80000000u >> pos)) : cur;
      }
      left -= 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* r = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds.rec) + at[e]);
        pos4[e] = make_float3(r[0], r[1], r[2]);
      }
    };
    // A trip's positions are requested during the trip before (right after that trip's own have been used: they take
    // the same registers), its colours at its start (they are needed at its end): no trip waits for LDS.
    bool more = __ballot(left > 0) != 0ull;
    if (more) take_four();
    while (more) {
      Light four[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* r = reinterpret_cast<const float*>(reinterpret_cast<const char*>(lds.rec) + at[e]);
        four[e] = Light{pos4[e].x, pos4[e].y, pos4[e].z, r[3], r[4], r[5]};
      }
      more = __ballot(left > 0) != 0ull;  // lights left after this trip, on any lane
      shade4<false, false, 4, decltype(take_four)&, void (*)(), true>(s, s, four, kc, probe, take_four, +[] {});
    }
    if (kChunked && !last_chunk) {
#pragma unroll
      for (int i = 0; i < 3; ++i) lds.sums[i][id] = s.s1[i], lds.sums[3 + i][id] = s.s2[i];
    }
    if (last_chunk) {
      float rgb[3];
      finish_pixel(s, alb, mat, true, rgb);
      if (kIs8Bit) {
        lds.out[id] = pack_unorm8<kOutMode>(rgb);
      } else if (inside) {
        uint8_t* orow = p.out + (size_t)y * p.pitch_out;
        if (kOutMode == kOutRgba16f)
          reinterpret_cast<uint2*>(orow)[x] = half_texel(rgb[0], rgb[1], rgb[2]);
        else
          reinterpret_cast<float4*>(orow)[x] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
      }
    }
  }
  }  // chunks
#ifdef VKR_X_TIMES
  if (lane == 0u) {
    const uint32_t wv = (blockIdx.y * gridDim.x + blockIdx.x) * kWaves + wave;
    if (wv < 65536u) {
      uint32_t hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_wave_times[4u * wv + 0u] = rt_start;
      g_wave_times[4u * wv + 1u] = wall_clock64();
      g_wave_times[4u * wv + 2u] = ((unsigned long long)xcc << 32) | hw;
      g_wave_times[4u * wv + 3u] = rt_lists;
    }
  }
#endif
  if (kIs8Bit) {
    __syncthreads();
    const uint32_t xl = (t & 15u) * kWalkHome, yl = t >> 4;
    const uint32_t x0 = tile_x * 32u + xl, y = tile_y * kWalkRows + yl;
    if (y < p.height && x0 < p.width) {
      uint32_t* o = reinterpret_cast<uint32_t*>(p.out + (size_t)y * p.pitch_out) + x0;
      const uint32_t* r = &lds.out[yl * 32u + xl];
      if (x0 + 1u < p.width)
        *reinterpret_cast<uint2*>(o) = make_uint2(r[0], r[1]);
      else
        o[0] = r[0];
    }
  }
}

// ------------------------------------------------------------------ bounds of a tile's pixel positions
// Wave-uniform bounds of a tile's pixel positions: a padded world-space box and the four side faces of the
// tile's view pyramid (outward unit normal + offset), plus reach = radius + pad.
struct TileBounds {
  float lo[3], hi[3], face[4][4], reach;
  float span;  // distance between the near and the far point of the box's first corner ray
};

// How far a pixel position as the resolve reconstructs it (deferred.comp:99-103 in binary32) can lie from the exact
// image of its clip-space point.  w = M3 cx + M7 cy + M11 d + M15 cancels as depth -> 1 (with the reference's
// near 0.1 / far 10000 the two last terms are about +-5 while their sum is 5e-4 at 1800 units), so each coordinate
// x / w carries a relative error of eps * (sum of |terms of w|) / |w| plus that of its own numerator; `terms_w` and
// `terms_xyz` bound the sums of absolute terms.  kPositionEps allows for every rounding of the evaluation (four
// products, three sums, the correctly rounded divisions) and doubles the result.
constexpr float kPositionEps = 0x1p-20f;  // 16 * 2^-24
__device__ __forceinline__ float position_error_bound(float max_abs_coord, float terms_w, float terms_xyz, float min_abs_w) {
  return kPositionEps * fma_(max_abs_coord, terms_w, terms_xyz) / min_abs_w + kPositionEps * max_abs_coord;
}

__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

// One wave computes the bounds of tile (tile_x, tile_y) from its depth range.  The pixel positions are the
// images of points of the clip-space box [xa,xb] x [ya,yb] x [dmin,dmax] under the projective map inverseVP;
// with w > 0 on all eight corners their hull contains them all.
__device__ __forceinline__ TileBounds tile_bounds(const DeferredParams& p, uint32_t tile_x, uint32_t tile_y, float dmin, float dmax, uint32_t lane) {
  const float xa = (float)(tile_x * 32u) / (float)p.width * 2.0f - 1.0f;
  const float xb = (float)min(tile_x * 32u + 32u, p.width) / (float)p.width * 2.0f - 1.0f;
  const float ya = (float)(tile_y * 32u + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
  const float yb = (float)(min(tile_y * 32u + 32u, p.height) + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
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
  // The pad covers the error of the reconstructed pixel positions.  Every term of the bound is largest at a corner
  // of the clip-space box (|linear form| and 1 / |w| of a linear w that keeps its sign are), so the maxima over the
  // eight corners bound it for every pixel of the tile.
  float terms_w = fabsf(M[3] * pcx) + fabsf(M[7] * pcy) + fabsf(M[11] * pcz) + fabsf(M[15]);
  float terms_xyz = fmaxf(fmaxf(fabsf(M[0] * pcx) + fabsf(M[4] * pcy) + fabsf(M[8] * pcz) + fabsf(M[12]),
                                fabsf(M[1] * pcx) + fabsf(M[5] * pcy) + fabsf(M[9] * pcz) + fabsf(M[13])),
                          fabsf(M[2] * pcx) + fabsf(M[6] * pcy) + fabsf(M[10] * pcz) + fabsf(M[14]));
  float min_w = fabsf(ww);
#pragma unroll
  for (int off = 1; off <= 4; off <<= 1) {
    terms_w = fmaxf(terms_w, __shfl_xor(terms_w, off));
    terms_xyz = fmaxf(terms_xyz, __shfl_xor(terms_xyz, off));
    min_w = fminf(min_w, __shfl_xor(min_w, off));
  }
  const float pad = position_error_bound(scale, terms_w, terms_xyz, min_w) + 1e-3f;
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
  // signed distance of x to the face = dot(n, x) - dot(n, near point c)
  const float fw = -(fx * q[0] + fy * q[1] + fz * q[2]);
  TileBounds b;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    b.lo[a] = sane ? lane_value(box_lo[a], 0) - lane_value(pad, 0) : -INFINITY;
    b.hi[a] = sane ? lane_value(box_hi[a], 0) + lane_value(pad, 0) : INFINITY;
  }
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    b.face[f][0] = lane_value(fx, f);
    b.face[f][1] = lane_value(fy, f);
    b.face[f][2] = lane_value(fz, f);
    b.face[f][3] = lane_value(fw, f);
  }
  b.reach = sqrt_(p.radius2) + lane_value(pad, 0);
  const float sx = lane_value(q[0], 4) - lane_value(q[0], 0), sy = lane_value(q[1], 4) - lane_value(q[1], 0), sz = lane_value(q[2], 4) - lane_value(q[2], 0);
  b.span = sane ? sqrt_(fma_(sz, sz, fma_(sy, sy, sx * sx))) : 0.0f;
  return b;
}

// ------------------------------------------------------------------ shading from tile light lists (radius cutoff)
// Strips: every wave works on its own (four to a workgroup, sharing nothing - see deferred_lists_kernel); the unit of
// work is a 32 x 4 pixel strip of a lit tile: lane l owns the two pixels
// (2*(l&15), 2*(l&15)+1) of row l>>4 of the strip, shaded as one pair, two lights per loop trip (the same four
// evaluations per trip as the whole-table kernel).  A lit tile is 8 units.  What bounds this launch on sparse
// frames is not arithmetic but the chain of dependent memory round trips in front of every unit (queue entry ->
// list length -> light indices -> light records), a few microseconds each time: with 256-thread workgroups
// taking whole or half tiles, a CU had 3-4 such chains in flight and the launch took (units per workgroup) x
// (chain), 34 us for 690 lit tiles of at most 6 lights.  Independent waves put 16 chains in flight per CU,
// need no barriers, and their list chunk of 64 lights is one record per lane.
// Units are dealt out statically: wave w of the grid takes units w, w + G, ... of the lit tiles in queue order,
// heaviest list class first (the eight strips of a tile land on neighbouring waves, so its list and its plane
// rows are fetched together).  The queue lengths are fixed before the launch, so every wave's loop ends.
// A tile whose list overflowed its capacity walks its coarse bin's candidates (or the whole table) instead; the
// per-pixel cutoff makes any superset of the reaching lights give the same bits.
// Measured and dropped: asking for the list's length, the lane's first list entry (before the length is known) and
// the planes together, so that the light records are requested one round trip earlier - 37.8 / 86.0 / 172.8 us
// against 38.3 / 86.2 / 170.1 us (4K x 256, 4K x 4096, 8K x 4096): with 16 waves per CU in different phases the
// chain's length is not what the launch waits for any more.
constexpr uint32_t kStripRows = 4, kStripsPerTile = 32 / kStripRows, kWaveChunk = 64;
#ifndef VKR_X_DEEP_MIN
#define VKR_X_DEEP_MIN 8
#endif
constexpr uint32_t kDeepMinList = VKR_X_DEEP_MIN;  // shorter lists are walked whole: narrowing them costs more than it saves
constexpr float kDeepSpanRadii = 1.0f;    // a tile is deep when its near and far ends are further apart than this many radii

// LDS of one wave in the strips role
struct StripLds {
  float s_light[(2 * kWaveChunk + 4) * 4];  // float4 records; +1 pair: the prefetch of the last iteration reads past the list
  uint32_t s_keep[4][64];                   // albedo and material words, waiting for the end of the loop
};

// The strips role of deferred_lists_kernel: this wave takes units unit_first, unit_first + unit_stride, ...
template <int kOutMode>
__device__ __forceinline__ void shade_strips(const DeferredParams& p, StripLds& lds, uint32_t unit_first, uint32_t unit_stride) {
  float4* const s_light = reinterpret_cast<float4*>(lds.s_light);
  uint32_t (&s_keep)[4][64] = lds.s_keep;
  constexpr bool kIs8Bit = kOutMode == kOutRgba8 || kOutMode == kOutBgra8;
  const uint32_t t = threadIdx.x & 63u;
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  const uint32_t L = p.light_count;
  LoopConsts kc;
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.radius2_scaled) : "s"(p.radius2 * kRadiusStepScale));

  uint32_t class_end[kListClasses];
  {
    uint32_t run = 0;
#pragma unroll
    for (int c = kListClasses - 1; c >= 0; --c) {
      run += p.lit_counters[c];
      class_end[kListClasses - 1 - c] = run;
    }
  }
  const uint32_t n_tiles = p.tiles_x * p.tiles_y;
  const uint32_t q_end = kStripsPerTile * class_end[kListClasses - 1];
#pragma unroll 1
  for (uint32_t q = unit_first; q < q_end; q += unit_stride) {
    const uint32_t pos = q / kStripsPerTile, strip = q % kStripsPerTile;
    uint32_t cls = kListClasses - 1u, first = 0u;  // position of the concatenated queues -> (class, index)
#pragma unroll
    for (uint32_t k = 0; k + 1u < kListClasses; ++k)
      if (pos >= class_end[k]) cls = kListClasses - 2u - k, first = class_end[k];
    const uint32_t tile = p.lit_queue[(size_t)cls * n_tiles + (pos - first)];
    const uint32_t tile_y = tile / p.tiles_x, tile_x = tile - tile_y * p.tiles_x;
    const uint32_t word = p.tile_count[tile];
    uint32_t count = L;                           // lights this unit walks
    const uint16_t* __restrict__ tl = nullptr;    // the tile's own list, or
    const uint32_t* __restrict__ cand = nullptr;  // its bin's candidates when the list overflowed
    if (word & kTileOverflow) {
      if (p.bin_list) {
        const uint32_t bin_index = (tile_y * 32u / kBinPixels) * p.bins_x + tile_x * 32u / kBinPixels;
        cand = p.bin_list + (size_t)bin_index * L;
        count = p.bin_count[bin_index];
      }
    } else {
      tl = p.tile_list + (size_t)tile * p.list_cap;
      count = word & kTileCountMask;
    }
    // Where the lane's two pixels are (neighbours in a row): derived from the thread index where needed (through a copy
    // the compiler cannot see through) rather than kept across the light loop, which leaves no register for values that
    // wait.  A pair beyond the image is replaced by the tile's first (in bounds by construction) and not stored.
    struct Where {
      uint32_t x0, y, lx, ly;
      bool inside;
    };
    auto where = [&](uint32_t tq) {
      uint32_t ox = tile_x * 32u, oy = tile_y * 32u, oys = tile_y * 32u + strip * kStripRows;
      asm volatile("" : "+v"(tq), "+s"(ox), "+s"(oy), "+s"(oys));
      Where w;
      w.x0 = ox + (tq & 15u) * 2u;
      w.y = oys + (tq >> 4);
      w.inside = (w.x0 < p.width) && (w.y < p.height);
      w.lx = w.inside ? w.x0 : ox;
      w.ly = w.inside ? w.y : oy;
      return w;
    };
    const Where w0 = where(t);
    if (__ballot(w0.inside) == 0ull) continue;  // a strip below the image's last row (wave-uniform)
    const uint2 alb2 = *reinterpret_cast<const uint2*>(p.albedo + (size_t)w0.ly * p.pitch_albedo + (size_t)w0.lx * 4u);
    const uint2 mat2 = *reinterpret_cast<const uint2*>(p.material + (size_t)w0.ly * p.pitch_material + (size_t)w0.lx * 4u);
    const float2 dep2 = *reinterpret_cast<const float2*>(p.depth + (size_t)w0.ly * p.pitch_depth + (size_t)w0.lx * 4u);
    const uint4 nrm = *reinterpret_cast<const uint4*>(p.normal + (size_t)w0.ly * p.pitch_normal + (size_t)w0.lx * 8u);
    const float4 black_pos = make_float4(0.0f, 0.0f, 1e18f, 0.0f), black_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    // A chunk of the list is one index, then one record, per lane.  The first chunk's records are requested here,
    // alongside the planes; of a later chunk only the index is requested ahead (one register across the light loop
    // where the record would take eight), the record when the chunk's turn comes.
    auto fetch_index = [&](uint32_t base) -> uint32_t {
      uint32_t tq = t;
      asm volatile("" : "+v"(tq));  // the addresses below are derived per call, not kept in registers across the light loop
      const uint32_t j = min(base + tq, count - 1u);
      return tl ? (uint32_t)tl[j] : (cand ? cand[j] : j);
    };
    auto fetch_records = [&](uint32_t base, uint32_t idx, float4& lp, float4& lc) {
      lp = black_pos;
      lc = black_col;
      if (base + t < count) {
        lp = lights[3u * idx];       // pos.xyz, opening_angle (unused)
        lc = lights[3u * idx + 1u];  // color.xyz, penumbra_angle (unused)
      }
    };
    float4 lp = black_pos, lc = black_col;
    if (count != 0u) fetch_records(0u, fetch_index(0u), lp, lc);
    uint32_t width = p.width, full_height = p.full_height;
    asm volatile("" : "+s"(width), "+s"(full_height));  // their reciprocals are derived per unit, not kept in vector registers
    const ExactRcp by_width = exact_rcp((float)width), by_height = exact_rcp((float)full_height);
    const float cy = clip_coord(w0.ly + p.row_offset, by_height);
    Pixel s[2];
    setup_pixel(s[0], p, clip_coord(w0.lx, by_width), cy, dep2.x, mat2.x, nrm.x, nrm.y);
    setup_pixel(s[1], p, clip_coord(w0.lx + 1u, by_width), cy, dep2.y, mat2.y, nrm.z, nrm.w);
    // the albedo and material words are decoded after the loop: they wait in LDS, one column per lane
    s_keep[0][t] = alb2.x;
    s_keep[1][t] = alb2.y;
    s_keep[2][t] = mat2.x;
    s_keep[3][t] = mat2.y;

    uint32_t probe_left = kProbeBudget;
    // a wave whose pixels all have a v.n that is a number <= 1e-6 gets 0 from every light (deferred.comp:43,84) and
    // does not need to know whether one lay in range: it keeps its ambient colour either way
    if (__ballot(!(s[0].vn <= kEarlyOut) || !(s[1].vn <= kEarlyOut)) == 0ull) {
      VKR_COUNT_TRIPS((count + 1u) / 2u, 0u, (count + 1u) / 2u);
      count = 0u;
    }
    uint32_t next_idx = 0;
    for (uint32_t base = 0; base < count; base += kWaveChunk) {
      const uint32_t n = min(kWaveChunk, count - base);
      const uint32_t n2 = (n + 1u) & ~1u;
      // one wave: its LDS accesses execute in program order, no barrier is needed between the loop's reads of
      // the previous chunk and these writes
      if (base != 0u) fetch_records(base, next_idx, lp, lc);
      s_light[2u * t] = lp;
      s_light[2u * t + 1u] = lc;
      Probe probe{probe_budget_for(n2 / 2u, probe_left)};
      if (base + kWaveChunk < count) next_idx = fetch_index(base + kWaveChunk);  // the next chunk's indices, while this one is shaded
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float4* rec = s_light;
      float4 pa = rec[0], ca = rec[1], pb = rec[2], cb = rec[3];
      uint32_t j = 0;
      for (; j < n2 && probe.budget != 0u; j += 2u) {  // see deferred_kernel: the loop with the test, then the one without
        const Light AB[2] = {{pa.x, pa.y, pa.z, ca.x, ca.y, ca.z}, {pb.x, pb.y, pb.z, cb.x, cb.y, cb.z}};
        rec += 4;  // next pair (one pair beyond the list on the last trip: read, never used)
        shade4<true, true>(
            s[0], s[1], AB, kc, probe, [&] { pa = rec[0], pb = rec[2]; }, [&] { ca = rec[1], cb = rec[3]; });
      }
      if (j < n2) {
        const float4* rec2 = s_light + 2u * j;
        float4 pa2 = rec2[0], ca2 = rec2[1], pb2 = rec2[2], cb2 = rec2[3];
        for (; j < n2; j += 2u) {
          const Light AB[2] = {{pa2.x, pa2.y, pa2.z, ca2.x, ca2.y, ca2.z}, {pb2.x, pb2.y, pb2.z, cb2.x, cb2.y, cb2.z}};
          rec2 += 4;
          pa2 = rec2[0];
          ca2 = rec2[1];
          pb2 = rec2[2];
          cb2 = rec2[3];
          shade4<true, false>(s[0], s[1], AB, kc, probe, [] {}, [] {});
        }
      }
      probe_left = probe.budget;
#ifdef VKR_X_COUNT
      VKR_COUNT_TRIPS(n2 / 2u, probe.skipped, 0u);
#endif
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    float rgb[2][3];
    finish_pixel(s[0], s_keep[0][t], s_keep[2][t], s[0].in_range > 0.0f, rgb[0]);
    finish_pixel(s[1], s_keep[1][t], s_keep[3][t], s[1].in_range > 0.0f, rgb[1]);
    const Where w1 = where(t);
    if (w1.inside) {
      const uint32_t x0 = w1.x0, y = w1.y;
      const bool both = x0 + 1u < p.width;
      uint8_t* orow = p.out + (size_t)y * p.pitch_out;
      if (kIs8Bit) {
        uint32_t* o = reinterpret_cast<uint32_t*>(orow) + x0;
        const uint32_t t0 = pack_unorm8<kOutMode>(rgb[0]), t1 = pack_unorm8<kOutMode>(rgb[1]);
        if (both)
          *reinterpret_cast<uint2*>(o) = make_uint2(t0, t1);
        else
          o[0] = t0;
      } else if (kOutMode == kOutRgba16f) {
        uint2* o = reinterpret_cast<uint2*>(orow) + x0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i == 0 || both) o[i] = half_texel(rgb[i][0], rgb[i][1], rgb[i][2]);
        }
      } else {
        float4* o = reinterpret_cast<float4*>(orow) + x0;
        o[0] = make_float4(rgb[0][0], rgb[0][1], rgb[0][2], 1.0f);
        if (both) o[1] = make_float4(rgb[1][0], rgb[1][1], rgb[1][2], 1.0f);
      }
    }
  }
}

// The tiles that tile_lists_kernel marked deep (more depth than a light reaches).  A workgroup of four waves takes one
// such tile.  The tile's four planes go through LDS (read from memory once, in whole rows); the tile's pixels, in the
// depth order tile_lists_kernel wrote, form 16 units of 64 - one pixel per lane, wherever it lies in the tile - and
// units 2g and 2g + 1 walk the list of depth group g, four lights per trip (the same four evaluations per trip, and
// per pixel the same sums in the same order, as everywhere else).  The waves take the units one at a time, the
// farthest first: clip-space depth crowds the far distances together, so the last groups of the depth order span the
// most space and keep the longest lists (uncorrelated depth at 8K x 4096: 22 lights and more against 1-3) - begun
// first, they end with the rest.  The 8-bit result leaves through LDS again, in whole rows.
// Measured on the way (8K x 4096, depth uncorrelated between neighbours, this kernel alone): pixels fetched straight
// from memory in depth order, no LDS: 1.2 ms (every wave touches nearly every cache line of the tile, in 4-byte
// pieces); planes through LDS, wave w shading groups w and w + 4 as pairs of pixels: 0.53 ms, of which a workgroup
// spends 7 us of its 16.5 us per tile waiting for the wave that has the last group (10 us against 3 us for any other
// group); the same with everything a wave needs for both of its groups requested ahead of the tile's barrier: no
// different (0.81 against 0.82 ms for the call); groups taken from a counter, farthest first: 14.6 us per tile.
// In the present form, also tried: the records of each group's first 16 lights fetched with the planes, so that a unit
// with a short list starts without a round trip of its own - slower on every frame (0.676 against 0.658 ms here, 0.146
// against 0.139 ms on the analytic 8K frame): the second round trip ahead of the tile's barrier stalls all four waves,
// while a unit's own fetch is covered by the other waves' work.
// LDS of a workgroup in the deep-tile role
struct DeepLds {
  float s_light_of[4][(2 * kWaveChunk + 8) * 4];  // float4 records per wave; + one trip: the prefetch of the last trip reads past the list
  uint32_t s_alb[1024], s_mat[1024];  // the tile's planes, texel y * 32 + x; a pixel's 8-bit result replaces its albedo word
  float s_dep[1024];
  uint32_t s_nrm[1024][2];
  // what every unit would otherwise fetch for itself, one round trip after the other, before it can ask for its first
  // light records: the tile's depth order, the groups' list lengths and the first 64 entries of each group's list
  uint16_t s_order[1024], s_entry[kStripsPerTile][kWaveChunk];
  uint32_t s_length[kStripsPerTile];
  uint32_t s_next_unit;  // how many of the tile's units have been taken
};

// The deep-tile role of deferred_lists_kernel: this workgroup takes tiles tile_first, tile_first + tile_stride, ... of the deep queue
template <int kOutMode>
__device__ __forceinline__ void shade_deep_tiles(const DeferredParams& p, DeepLds& lds, uint32_t tile_first, uint32_t tile_stride) {
  constexpr uint32_t kUnits = 16;
  uint32_t* const s_alb = lds.s_alb;
  uint32_t* const s_mat = lds.s_mat;
  uint32_t* const s_out = lds.s_alb;
  float* const s_dep = lds.s_dep;
  uint2* const s_nrm = reinterpret_cast<uint2*>(lds.s_nrm);
  uint16_t* const s_order = lds.s_order;
  uint16_t (&s_entry)[kStripsPerTile][kWaveChunk] = lds.s_entry;
  uint32_t* const s_length = lds.s_length;
  uint32_t& s_next_unit = lds.s_next_unit;
  constexpr bool kIs8Bit = kOutMode == kOutRgba8 || kOutMode == kOutBgra8;
  const uint32_t t = threadIdx.x & 63u, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // lane, wave of the workgroup (the same in every lane)
  float4* const s_light = reinterpret_cast<float4*>(lds.s_light_of[wv]);
  const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
  LoopConsts kc;
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_scale) : "s"(kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.step_bias) : "s"(-kEarlyOut * kStepScale));
  asm volatile("v_mov_b32 %0, %1" : "=v"(kc.radius2_scaled) : "s"(p.radius2 * kRadiusStepScale));
  const uint32_t n_tiles = p.tiles_x * p.tiles_y;
  const uint32_t q_end = p.lit_counters[kListClasses];
#pragma unroll 1
  for (uint32_t q = tile_first; q < q_end; q += tile_stride) {
    const uint32_t tile = p.lit_queue[(size_t)kListClasses * n_tiles + q];
    const uint32_t tile_y = tile / p.tiles_x, tile_x = tile - tile_y * p.tiles_x;
    const uint32_t word = p.tile_count[tile];
#ifdef VKR_X_TIMES  // measurement builds only (tools/tile_phase_times.py deep): the wall clock at the end of each phase of a deep tile's workgroup
    unsigned long long dstamp[8] = {};
#define VKR_DSTAMP(i) dstamp[i] = wall_clock64()
    VKR_DSTAMP(0);
#else
#define VKR_DSTAMP(i)
#endif
    {
      // the tile's planes, 16 bytes per thread and request (rows hold whole 4-texel groups): 256 requests each for albedo,
      // material and depth, 512 for the normals
      uint32_t c = threadIdx.x;
      asm volatile("" : "+v"(c));  // addresses derived per tile, not kept from the kernel's entry across the light loop
      const uint32_t r = c >> 3, x4 = (c & 7u) * 4u;
      const uint32_t gx = tile_x * 32u + x4, gy = tile_y * 32u + r;
      if (gx < p.width && gy < p.height) {
        *reinterpret_cast<uint4*>(s_alb + r * 32u + x4) = *reinterpret_cast<const uint4*>(p.albedo + (size_t)gy * p.pitch_albedo + (size_t)gx * 4u);
        *reinterpret_cast<uint4*>(s_mat + r * 32u + x4) = *reinterpret_cast<const uint4*>(p.material + (size_t)gy * p.pitch_material + (size_t)gx * 4u);
        *reinterpret_cast<float4*>(s_dep + r * 32u + x4) = *reinterpret_cast<const float4*>(p.depth + (size_t)gy * p.pitch_depth + (size_t)gx * 4u);
      }
#pragma unroll
      for (uint32_t h = 0; h < 2u; ++h) {  // two 8-byte texels per request
        const uint32_t c2 = c + 256u * h, r2 = c2 >> 4, x2 = (c2 & 15u) * 2u;
        const uint32_t hx = tile_x * 32u + x2, hy = tile_y * 32u + r2;
        if (hx < p.width && hy < p.height)
          *reinterpret_cast<uint4*>(s_nrm + r2 * 32u + x2) = *reinterpret_cast<const uint4*>(p.normal + (size_t)hy * p.pitch_normal + (size_t)hx * 8u);
      }
      reinterpret_cast<uint2*>(s_order)[c] = reinterpret_cast<const uint2*>(p.tile_order + (size_t)tile * 1024u)[c];
      // (a list's memory is there whatever its length: entries beyond the length are read and never used)
      reinterpret_cast<uint32_t*>(&s_entry[0][0])[c] =
          reinterpret_cast<const uint32_t*>(p.group_list + ((size_t)tile * kStripsPerTile + (c >> 5)) * kGroupListCap)[c & 31u];
      if (c < kStripsPerTile) s_length[c] = p.group_count[(size_t)tile * kStripsPerTile + c];
      if (c == 0u) s_next_unit = 0u;
    }
    VKR_DSTAMP(1);  // plane requests issued, rows stored
    __syncthreads();
    VKR_DSTAMP(2);  // planes in LDS
    const uint32_t tile_pixels = min(32u, p.width - tile_x * 32u) * min(32u, p.height - tile_y * 32u);
#pragma unroll 1
    for (uint32_t turn = 0;; ++turn) {
      (void)turn;
      uint32_t taken = 0;
      if (t == 0u) taken = atomicAdd(&s_next_unit, 1u);
      taken = __builtin_amdgcn_readfirstlane(taken);
      if (taken >= kUnits) break;
      const uint32_t unit = kUnits - 1u - taken;
      if (unit * 64u >= tile_pixels) continue;  // a tile at the image's edge has fewer units (wave-uniform)
      // the list of the unit's depth group: the part of the tile's list that can reach the group's depth range
      // (tile_lists_kernel); when that outgrew its capacity, the tile's own list
      const uint32_t group = unit >> 1;
      uint32_t count = __builtin_amdgcn_readfirstlane(s_length[group]);
      const uint16_t* __restrict__ tl = p.group_list + ((size_t)tile * kStripsPerTile + group) * kGroupListCap;
      const bool own_list = count != kGroupOverflow;
      if (!own_list) {
        tl = p.tile_list + (size_t)tile * p.list_cap;
        count = word & kTileCountMask;
      }
      // The lane's pixel: entry unit * 64 + lane of the tile's depth order (id = y * 32 + x within the tile); beyond the
      // tile's last pixel, the unit's first, not stored.  Derived where needed, through a copy the compiler cannot see
      // through, rather than kept across the light loop.
      auto which = [&](uint32_t tq, bool& mine) -> uint32_t {
        uint32_t first = unit * 64u;
        asm volatile("" : "+v"(tq), "+s"(first));
        mine = first + tq < tile_pixels;
        return (uint32_t)s_order[first + (mine ? tq : 0u)];
      };
      bool mine0;
      const uint32_t id0 = which(t, mine0);
      const float4 black_pos = make_float4(0.0f, 0.0f, 1e18f, 0.0f), black_col = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      auto fetch_records = [&](uint32_t base, float4& lp, float4& lc) {  // one list entry, then its record, per lane
        uint32_t tq = t;
        asm volatile("" : "+v"(tq));
        lp = black_pos;
        lc = black_col;
        if (base + tq < count) {
          const uint32_t idx = (base == 0u && own_list) ? (uint32_t)s_entry[group][tq] : (uint32_t)tl[base + tq];
          lp = lights[3u * idx];       // pos.xyz, opening_angle (unused)
          lc = lights[3u * idx + 1u];  // color.xyz, penumbra_angle (unused)
        }
      };
      float4 lp, lc;
      fetch_records(0u, lp, lc);
      Pixel s;
      {
        uint32_t width = p.width, full_height = p.full_height;
        asm volatile("" : "+s"(width), "+s"(full_height));  // their reciprocals are derived per unit, not kept in vector registers
        const ExactRcp by_width = exact_rcp((float)width), by_height = exact_rcp((float)full_height);
        const uint2 n2 = s_nrm[id0];
        setup_pixel(s, p, clip_coord(tile_x * 32u + (id0 & 31u), by_width), clip_coord(tile_y * 32u + (id0 >> 5) + p.row_offset, by_height), s_dep[id0],
                    s_mat[id0], n2.x, n2.y);
      }
      uint32_t probe_left = kProbeBudget;
      // a wave whose pixels all have a v.n that is a number <= 1e-6 gets 0 from every light (deferred.comp:43,84) and
      // does not need to know whether one lay in range: it keeps its ambient colour either way
      if (__ballot(!(s.vn <= kEarlyOut)) == 0ull) {
        VKR_COUNT_TRIPS((count + 3u) / 4u, 0u, (count + 3u) / 4u);
        count = 0u;
      }
      for (uint32_t base = 0; base < count; base += kWaveChunk) {
        const uint32_t n = min(kWaveChunk, count - base);
        const uint32_t n4 = (n + 3u) & ~3u;  // padded to whole trips with black lights (lanes beyond n hold them)
        // one wave: its LDS accesses execute in program order, no barrier is needed between the loop's reads of
        // the previous chunk and these writes
        if (base != 0u) fetch_records(base, lp, lc);
        s_light[2u * t] = lp;
        s_light[2u * t + 1u] = lc;
        Probe probe{probe_budget_for(n4 / 4u, probe_left)};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float4* rec = s_light;
        float4 q0 = rec[0], c0 = rec[1], q1 = rec[2], c1 = rec[3], q2 = rec[4], c2 = rec[5], q3 = rec[6], c3 = rec[7];
        uint32_t j = 0;
        for (; j < n4 && probe.budget != 0u; j += 4u) {  // see deferred_kernel: the loop with the test, then the one without
          const Light four[4] = {{q0.x, q0.y, q0.z, c0.x, c0.y, c0.z}, {q1.x, q1.y, q1.z, c1.x, c1.y, c1.z},
                                 {q2.x, q2.y, q2.z, c2.x, c2.y, c2.z}, {q3.x, q3.y, q3.z, c3.x, c3.y, c3.z}};
          rec += 8;  // next trip (one trip beyond the list on the last one: read, never used)
          shade4<true, true>(
              s, s, four, kc, probe, [&] { q0 = rec[0], q1 = rec[2], q2 = rec[4], q3 = rec[6]; }, [&] { c0 = rec[1], c1 = rec[3], c2 = rec[5], c3 = rec[7]; });
        }
        if (j < n4) {
          uint32_t at = 2u * j;  // see deferred_pixel_kernel: one opaque address and eight immediate offsets per trip
          asm volatile("" : "+v"(at));
          const float4* rec2 = s_light + at;
          float4 r0 = rec2[0], d0 = rec2[1], r1 = rec2[2], d1 = rec2[3], r2 = rec2[4], d2 = rec2[5], r3 = rec2[6], d3 = rec2[7];
          for (; j < n4; j += 4u) {
            const Light four[4] = {{r0.x, r0.y, r0.z, d0.x, d0.y, d0.z}, {r1.x, r1.y, r1.z, d1.x, d1.y, d1.z},
                                   {r2.x, r2.y, r2.z, d2.x, d2.y, d2.z}, {r3.x, r3.y, r3.z, d3.x, d3.y, d3.z}};
            at += 8u;
            asm volatile("" : "+v"(at));
            rec2 = s_light + at;
            r0 = rec2[0], d0 = rec2[1], r1 = rec2[2], d1 = rec2[3], r2 = rec2[4], d2 = rec2[5], r3 = rec2[6], d3 = rec2[7];
            shade4<true, false>(s, s, four, kc, probe, [] {}, [] {});
          }
        }
        probe_left = probe.budget;
#ifdef VKR_X_COUNT
        VKR_COUNT_TRIPS(n4 / 4u, probe.skipped, 0u);
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      bool mine1;
      const uint32_t id1 = which(t, mine1);
      float rgb[3];
      finish_pixel(s, s_alb[id1], s_mat[id1], s.in_range > 0.0f, rgb);
      if (mine1) {
        if (kIs8Bit) {
          s_out[id1] = pack_unorm8<kOutMode>(rgb);  // leaves in whole rows, below
        } else {
          uint8_t* orow = p.out + (size_t)(tile_y * 32u + (id1 >> 5)) * p.pitch_out;
          const uint32_t x = tile_x * 32u + (id1 & 31u);
          if (kOutMode == kOutRgba16f)
            reinterpret_cast<uint2*>(orow)[x] = half_texel(rgb[0], rgb[1], rgb[2]);
          else
            reinterpret_cast<float4*>(orow)[x] = make_float4(rgb[0], rgb[1], rgb[2], 1.0f);
        }
      }
#ifdef VKR_X_TIMES
      dstamp[3u + min(turn, 1u)] = wall_clock64();  // this wave's first / last unit shaded
#endif
    }
    __syncthreads();  // every unit's results are in s_out, and nobody reads the planes any more
    VKR_DSTAMP(5);
    if (kIs8Bit) {
      uint32_t c = threadIdx.x;
      asm volatile("" : "+v"(c));
      const uint32_t r = c >> 3, x4 = (c & 7u) * 4u;
      const uint32_t gx = tile_x * 32u + x4, gy = tile_y * 32u + r;
      if (gx < p.width && gy < p.height) {
        const uint32_t packed[4] = {s_out[r * 32u + x4], s_out[r * 32u + x4 + 1u], s_out[r * 32u + x4 + 2u], s_out[r * 32u + x4 + 3u]};
        store_rgba8x4(p.out + (size_t)gy * p.pitch_out, gx, min(4u, p.width - gx), packed);
      }
    }
    __syncthreads();  // s_out and the planes are free for the workgroup's next tile
#ifdef VKR_X_TIMES
    VKR_DSTAMP(6);
    if (VKR_X_TIMES == 2 && t == 0u && tile < 8192u) {
#pragma unroll
      for (int i = 0; i < 7; ++i) g_wave_times[32u * tile + 8u * wv + i] = dstamp[i];
    }
#endif
  }
}

// One launch shades everything the tile kernel queued.  Workgroups of four waves; the first `deep_groups` workgroups of
// the grid take whole deep tiles, the others shade strips, each wave on its own.  A deep tile is the longer piece of
// work (12 us against 3-5 for a strip), so those workgroups go first, and strips workgroups take their places as they
// end: as two launches the two kinds ran one after the other, neither filling the device (8K x 4096 on the analytic
// scene: 24 + 37 us; 4K x 4096: 0.088 -> 0.079 ms for the call).  Interleaving the roles in launch order instead (one
// deep-tile workgroup in front of every four strips workgroups) was slower where every tile is deep (0.82 against
// 0.67 ms: the deep-tile workgroups reach the CUs unevenly).
template <int kOutMode>
__global__ __launch_bounds__(256, 4) void deferred_lists_kernel(const DeferredParams p, uint32_t deep_groups) {
  constexpr size_t kLdsBytes = sizeof(DeepLds) > 4 * sizeof(StripLds) ? sizeof(DeepLds) : 4 * sizeof(StripLds);
  __shared__ __align__(16) unsigned char lds_raw[kLdsBytes];
  // for the host's next call: how many tiles of this frame qualify as deep (a store to host memory, nobody waits for it)
  if (blockIdx.x == 0u && threadIdx.x == 0u && p.deep_hint) __hip_atomic_store(p.deep_hint, p.lit_counters[kDeepCounter], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (blockIdx.x < deep_groups) {
    shade_deep_tiles<kOutMode>(p, *reinterpret_cast<DeepLds*>(lds_raw), blockIdx.x, deep_groups);
  } else {
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    shade_strips<kOutMode>(p, reinterpret_cast<StripLds*>(lds_raw)[wv], (blockIdx.x - deep_groups) * 4u + wv, (gridDim.x - deep_groups) * 4u);
  }
}

// ------------------------------------------------------------------ tile light lists (radius cutoff)
// One wave per 32 x 32 pixel tile (four tiles per workgroup; nothing is shared between the waves but the queue
// push at the end).  Lane l owns the 4-pixel groups (4*(l&7) .. +3, (l>>3) + 8k), k = 0..3, of the tile: four
// 128-bit depth loads and four albedo loads in flight per lane.  The wave derives the tile's bounds from its
// depth range, tests the tile's candidates (the lights of its coarse bin when the launch carries bins, else the
// whole table) 64 at a time and writes the indices of those that can reach the tile, in table order, to the
// tile's list.  A tile that no light reaches is finished here: every light sum of deferred.comp:120-129 is
// zero for its pixels, whatever their positions and normals are, so they are their ambient term
// (deferred.comp:107) - 12 bytes per pixel instead of 24, and the shading kernel never sees the tile.  The other
// tiles are queued for deferred_lists_kernel.
template <int kOutMode>
__global__ __launch_bounds__(256, 5) void tile_lists_kernel(const DeferredParams p, const TileListOut o) {
  __shared__ uint32_t s_lit[4], s_class[4];
  __shared__ uint32_t s_cell[4][64];  // per wave: pixels per depth cell, then where each cell's pixels begin
  __shared__ uint16_t s_sorted_id[4][1024];  // per wave, deep tiles: the tile's pixels (y * 32 + x) in depth order,
  __shared__ float s_sorted_depth[4][1024];  // and their depths in that order
  __shared__ uint16_t s_tlist[4][kChunk];  // per wave: the tile's list as it is written out (kChunk = list_cap)
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  const uint32_t n_tiles = p.tiles_x * p.tiles_y;
  const uint32_t tile = blockIdx.x * 4u + wv;
  if (blockIdx.x == 0u && t < kListCounterStride) o.counters_next[t] = 0u;  // the counters of the launch after this one
  uint32_t lit_tile = 0xFFFFFFFFu, lit_class = 0u;
  bool qualifies = false;  // as a deep tile (wave-uniform)
#ifdef VKR_X_TIMES  // measurement builds only (tools/tile_phase_times.py): the wall clock at the end of each phase of a tile's wave
  unsigned long long stamp[8] = {};
#define VKR_STAMP(i) stamp[i] = wall_clock64()
  VKR_STAMP(0);
#else
#define VKR_STAMP(i)
#endif
  if (tile < n_tiles) {  // wave-uniform
    const uint32_t tile_y = tile / p.tiles_x, tile_x = tile - tile_y * p.tiles_x;
    const uint32_t x0 = tile_x * 32u + (lane & 7u) * 4u;
    float4 dep[4];
    uint4 alb[4];
    bool in[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t y = tile_y * 32u + (lane >> 3) + 8u * k;
      in[k] = (x0 < p.width) && (y < p.height);
      const uint32_t lx = in[k] ? x0 : tile_x * 32u, ly = in[k] ? y : tile_y * 32u;
      dep[k] = *reinterpret_cast<const float4*>(p.depth + (size_t)ly * p.pitch_depth + (size_t)lx * 4u);
      // read once, and only of use when the tile turns out dark: streaming load
      alb[k] = load_stream_u4(p.albedo + (size_t)ly * p.pitch_albedo + (size_t)lx * 4u);
    }
    // the first candidates' indices are asked for now, behind the planes: where they come from depends on the tile's place
    // only, and their round trip passes while the depth range and the bounds are worked out (8K x 4096: 0.145 -> 0.141 ms;
    // their positions as well: five registers spilled at the 96 that five waves per SIMD allow, and slower)
    const float4* __restrict__ lights = reinterpret_cast<const float4*>(p.lights);
    const uint32_t L = p.light_count;
    const uint32_t bin_index = (tile_y * 32u / kBinPixels) * p.bins_x + tile_x * 32u / kBinPixels;
    const uint32_t* __restrict__ bin = p.bin_list ? p.bin_list + (size_t)bin_index * L : nullptr;
    const uint32_t LL = bin ? p.bin_count[bin_index] : L;
    constexpr uint32_t kBatch = 4;  // groups of 64 candidates in flight
    uint32_t idx[kBatch];
    float4 lp[kBatch];
    if (LL != 0u) {
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t pos = min(u * 64u + lane, LL - 1u);
        idx[u] = bin ? bin[pos] : pos;
      }
    }
    // depth range over the whole 4-pixel groups, as the shading lanes read them (a group's padding pixels
    // included: they only widen the bounds)
    float dmin = INFINITY, dmax = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dmin = fminf(dmin, fminf(fminf(dep[k].x, dep[k].y), fminf(dep[k].z, dep[k].w)));
      dmax = fmaxf(dmax, fmaxf(fmaxf(dep[k].x, dep[k].y), fmaxf(dep[k].z, dep[k].w)));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      dmin = fminf(dmin, __shfl_xor(dmin, off));
      dmax = fmaxf(dmax, __shfl_xor(dmax, off));
    }
    VKR_STAMP(1);  // depth arrived, range known
    const TileBounds b = tile_bounds(p, tile_x, tile_y, dmin, dmax, lane);
    VKR_STAMP(2);  // bounds

    uint16_t* __restrict__ list = o.tile_list + (size_t)tile * p.list_cap;
    uint32_t count = 0;  // wave-uniform
    for (uint32_t base = 0; base < LL; base += 64u * kBatch) {
      if (base != 0u) {
#pragma unroll
        for (uint32_t u = 0; u < kBatch; ++u) {
          const uint32_t pos = min(base + u * 64u + lane, LL - 1u);
          idx[u] = bin ? bin[pos] : pos;
        }
      }
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) lp[u] = lights[3u * idx[u]];
#pragma unroll
      for (uint32_t u = 0; u < kBatch; ++u) {
        const uint32_t pos = base + u * 64u + lane;
        const float dx = fmaxf(fmaxf(b.lo[0] - lp[u].x, lp[u].x - b.hi[0]), 0.0f);
        const float dy = fmaxf(fmaxf(b.lo[1] - lp[u].y, lp[u].y - b.hi[1]), 0.0f);
        const float dz = fmaxf(fmaxf(b.lo[2] - lp[u].z, lp[u].z - b.hi[2]), 0.0f);
        bool hit = (pos < LL) & ((dx * dx + dy * dy + dz * dz) < p.radius2);  // NaN position: never contributes, never listed
        if (__ballot(hit) != 0ull) {  // the box rejects most lights of most groups: one uniform branch skips the face tests
#pragma unroll
          for (int f = 0; f < 4; ++f)  // outside a side face by more than the radius: cannot reach any pixel of the tile
            hit &= !(((b.face[f][0] * lp[u].x + b.face[f][1] * lp[u].y) + b.face[f][2] * lp[u].z) + b.face[f][3] > b.reach);
        }
        const uint64_t hits = __ballot(hit);
        if (hit) {
          const uint32_t k = count + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
          if (k < p.list_cap) {
            list[k] = (uint16_t)idx[u];
            s_tlist[wv][k] = (uint16_t)idx[u];
          }
        }
        count += (uint32_t)__popcll(hits);
      }
    }
    VKR_STAMP(3);  // the tile's list
    bool deep = false;  // wave-uniform
    if (count != 0u) {
      // A tile whose pixels span more depth than a light reaches (an object's edge in front of a far background, or
      // depth that jumps from pixel to pixel) has a list most of whose lights miss most of its pixels.  Such a tile is
      // shaded in groups of pixels of similar depth, each of which narrows the list to the lights that reach its own
      // depth range: its pixels are written out grouped by 64 depth cells (a counting sort; the order inside a cell
      // is whatever the LDS atomics give, which changes the grouping and nothing else - any list that holds the
      // lights reaching a pixel, in table order, gives that pixel the same bytes).
      qualifies = count >= kDeepMinList && count <= p.list_cap && b.span > kDeepSpanRadii * sqrt_(p.radius2) && dmax > dmin;
      deep = qualifies && (p.flags & kFlagNoDeepTiles) == 0u;
      if (deep) {
        s_cell[wv][lane] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float to_cell = 64.0f / (dmax - dmin);
        uint32_t rank[4][4];
  #pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d4[4] = {dep[k].x, dep[k].y, dep[k].z, dep[k].w};
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t cell = min(63u, (uint32_t)((d4[i] - dmin) * to_cell));  // a depth that is not a number lands in cell 0
            rank[k][i] = cell << 16;
            if (in[k] && x0 + (uint32_t)i < p.width) rank[k][i] |= atomicAdd(&s_cell[wv][cell], 1u);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t mine = s_cell[wv][lane];
        uint32_t incl = mine;
  #pragma unroll
        for (uint32_t dd = 1; dd < 64u; dd <<= 1) {
          const uint32_t up = __shfl_up(incl, dd);
          if (lane >= dd) incl += up;
        }
        __builtin_amdgcn_wave_barrier();
        s_cell[wv][lane] = incl - mine;  // where the cell's pixels begin
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // The pixels go to their places of the depth order in LDS first: from there lane l writes places 16 l .. 16 l + 15
        // out as 32 contiguous bytes (the wave: the tile's 2 KiB in one piece; scattered straight to memory they were
        // 1024 two-byte stores), and the same lane's sixteen depths are all of one group (128 = 8 lanes' worth), so a
        // group's depth range is a minimum and maximum over eight lanes - no atomics.
  #pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float d4[4] = {dep[k].x, dep[k].y, dep[k].z, dep[k].w};
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (in[k] && x0 + (uint32_t)i < p.width) {
              const uint32_t at = s_cell[wv][rank[k][i] >> 16] + (rank[k][i] & 0xFFFFu);
              s_sorted_id[wv][at] = (uint16_t)((((lane >> 3) + 8u * k) << 5) | ((lane & 7u) * 4u + (uint32_t)i));
              s_sorted_depth[wv][at] = d4[i];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t tile_pixels = min(32u, p.width - tile_x * 32u) * min(32u, p.height - tile_y * 32u);
        {
          const uint4* ids = reinterpret_cast<const uint4*>(&s_sorted_id[wv][16u * lane]);
          uint4* out_ids = reinterpret_cast<uint4*>(o.tile_order + (size_t)tile * 1024u + 16u * lane);
          if (16u * lane < tile_pixels) {  // (places beyond the tile's last pixel hold whatever LDS held: never read)
            out_ids[0] = ids[0];
            out_ids[1] = ids[1];
          }
        }
        VKR_STAMP(4);  // pixel order written
        // Depth range of each group of 128 consecutive pixels of that order; a depth that is not a number (such a
        // pixel is lit by nothing) takes no part, fminf / fmaxf drop it.
        float gmin = INFINITY, gmax = -INFINITY;
  #pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 d = *reinterpret_cast<const float4*>(&s_sorted_depth[wv][16u * lane + 4u * j]);
          const float dd[4] = {d.x, d.y, d.z, d.w};
  #pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (16u * lane + 4u * j + (uint32_t)i < tile_pixels) {
              gmin = fminf(gmin, dd[i]);
              gmax = fmaxf(gmax, dd[i]);
            }
          }
        }
  #pragma unroll
        for (int off = 1; off <= 4; off <<= 1) {  // over the group's eight lanes
          gmin = fminf(gmin, __shfl_xor(gmin, off));
          gmax = fmaxf(gmax, __shfl_xor(gmax, off));
        }
        // All eight groups' boxes at once: lane 8g + 2c + k unprojects corner c of the tile at group g's least (k = 0) or
        // greatest (k = 1) depth; the box of a group is the hull of its eight points plus the error pad of the positions
        // (the tile's side faces do not depend on depth: the tile's list has passed them already).
        float glo[3], ghi[3];
        {
          const uint32_t c = (lane >> 1) & 3u;
          const float gd = (lane & 1u) ? gmax : gmin;  // (every lane of 8g .. 8g + 7 holds group g's range)
          const float xa = (float)(tile_x * 32u) / (float)p.width * 2.0f - 1.0f;
          const float xb = (float)min(tile_x * 32u + 32u, p.width) / (float)p.width * 2.0f - 1.0f;
          const float ya = (float)(tile_y * 32u + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
          const float yb = (float)(min(tile_y * 32u + 32u, p.height) + p.row_offset) / (float)p.full_height * 2.0f - 1.0f;
          const float pcx = (c == 1u || c == 2u) ? xb : xa, pcy = (c >= 2u) ? yb : ya;
          const float* M = p.inv_vp;
          const float wx = ((M[0] * pcx + M[4] * pcy) + M[8] * gd) + M[12];
          const float wy = ((M[1] * pcx + M[5] * pcy) + M[9] * gd) + M[13];
          const float wz = ((M[2] * pcx + M[6] * pcy) + M[10] * gd) + M[14];
          const float ww = ((M[3] * pcx + M[7] * pcy) + M[11] * gd) + M[15];
          const float q[3] = {wx / ww, wy / ww, wz / ww};
          bool ok = (ww > 1e-30f) && (fabsf(q[0]) < 1e30f) && (fabsf(q[1]) < 1e30f) && (fabsf(q[2]) < 1e30f);
          float terms_w = fabsf(M[3] * pcx) + fabsf(M[7] * pcy) + fabsf(M[11] * gd) + fabsf(M[15]);
          float terms_xyz = fmaxf(fmaxf(fabsf(M[0] * pcx) + fabsf(M[4] * pcy) + fabsf(M[8] * gd) + fabsf(M[12]),
                                        fabsf(M[1] * pcx) + fabsf(M[5] * pcy) + fabsf(M[9] * gd) + fabsf(M[13])),
                                  fabsf(M[2] * pcx) + fabsf(M[6] * pcy) + fabsf(M[10] * gd) + fabsf(M[14]));
          float min_w = fabsf(ww), scale = 0.0f;
  #pragma unroll
          for (int a = 0; a < 3; ++a) glo[a] = ghi[a] = q[a];
  #pragma unroll
          for (int off = 1; off <= 4; off <<= 1) {  // over the group's eight lanes
  #pragma unroll
            for (int a = 0; a < 3; ++a) {
              glo[a] = fminf(glo[a], __shfl_xor(glo[a], off));
              ghi[a] = fmaxf(ghi[a], __shfl_xor(ghi[a], off));
            }
            terms_w = fmaxf(terms_w, __shfl_xor(terms_w, off));
            terms_xyz = fmaxf(terms_xyz, __shfl_xor(terms_xyz, off));
            min_w = fminf(min_w, __shfl_xor(min_w, off));
            ok = ok && (__shfl_xor(ok ? 1 : 0, off) != 0);
          }
  #pragma unroll
          for (int a = 0; a < 3; ++a) scale = fmaxf(scale, fmaxf(fabsf(glo[a]), fabsf(ghi[a])));
          const float pad = position_error_bound(scale, terms_w, terms_xyz, min_w) + 1e-3f;
  #pragma unroll
          for (int a = 0; a < 3; ++a) {
            glo[a] = ok ? glo[a] - pad : -INFINITY;  // bounds that make no sense (a depth range of not-numbers, w <= 0): any light passes
            ghi[a] = ok ? ghi[a] + pad : INFINITY;
          }
        }
        VKR_STAMP(5);  // group ranges and boxes
        const uint32_t groups = (tile_pixels + 127u) / 128u;
        uint32_t kept[8] = {};  // wave-uniform
        for (uint32_t base = 0; base < count; base += 64u) {
          const uint32_t j = base + lane;
          const uint32_t idx = s_tlist[wv][min(j, count - 1u)];
          const float4 q = lights[3u * idx];
  #pragma unroll
          for (uint32_t g = 0; g < 8u; ++g) {
            if (g >= groups) break;  // wave-uniform
            const float lx = lane_value(glo[0], 8 * g), ly = lane_value(glo[1], 8 * g), lz = lane_value(glo[2], 8 * g);
            const float hx = lane_value(ghi[0], 8 * g), hy = lane_value(ghi[1], 8 * g), hz = lane_value(ghi[2], 8 * g);
            const float dx = fmaxf(fmaxf(lx - q.x, q.x - hx), 0.0f);
            const float dy = fmaxf(fmaxf(ly - q.y, q.y - hy), 0.0f);
            const float dz = fmaxf(fmaxf(lz - q.z, q.z - hz), 0.0f);
            const bool hit = (j < count) & ((dx * dx + dy * dy + dz * dz) < p.radius2);
            const uint64_t hits = __ballot(hit);
            if (hit) {
              const uint32_t k = kept[g] + __builtin_amdgcn_mbcnt_hi((uint32_t)(hits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hits, 0u));
              if (k < kGroupListCap) o.group_list[((size_t)tile * kStripsPerTile + g) * kGroupListCap + k] = (uint16_t)idx;
            }
            kept[g] += (uint32_t)__popcll(hits);
          }
        }
        if (lane < groups) {
          uint32_t mine = 0;
  #pragma unroll
          for (uint32_t g = 0; g < 8u; ++g) mine = lane == g ? kept[g] : mine;
          o.group_count[(size_t)tile * kStripsPerTile + lane] = (uint16_t)(mine > kGroupListCap ? kGroupOverflow : mine);
        }
      }
      lit_tile = tile;
      lit_class = deep ? kListClasses : (count > p.list_cap ? kListClasses - 1u : list_class(count));
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!in[k]) continue;
        const uint32_t y = tile_y * 32u + (lane >> 3) + 8u * k;
        const uint32_t valid = min(4u, p.width - x0);
        uint8_t* orow = p.out + (size_t)y * p.pitch_out;
        const uint32_t words[4] = {alb[k].x, alb[k].y, alb[k].z, alb[k].w};
        float rgb[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int ch = 0; ch < 3; ++ch) rgb[i][ch] = kAmbient * unorm8(words[i], ch);
        if (kOutMode == kOutRgba8 || kOutMode == kOutBgra8) {
          uint32_t packed[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) packed[i] = pack_unorm8<kOutMode>(rgb[i]);
          if (valid == 4u) {  // streaming store: measured 77 -> 67 us for this kernel on the 8K frame (90 % dark tiles)
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v4u v = {packed[0], packed[1], packed[2], packed[3]};
            __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(reinterpret_cast<uint32_t*>(orow) + x0));
          } else {
            store_rgba8x4(orow, x0, valid, packed);
          }
        } else {
          store_float4x4<kOutMode>(orow, x0, valid, rgb);
        }
      }
    }
    if (lane == 0u) o.tile_count[tile] = count > p.list_cap ? (count | kTileOverflow) : (deep ? (count | kTileDeep) : count);
    VKR_STAMP(6);  // group lists (or the dark tile's pixels) written
  }
  if (lane == 0u) {
    s_lit[wv] = lit_tile;
    s_class[wv] = lit_class | (qualifies ? 0x100u : 0u);
  }
  __syncthreads();
  if (t == kDeepCounter) {  // how many tiles qualify as deep, whether or not this launch treats them so
    uint32_t n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) n += (s_class[w] >> 8) & 1u;
    if (n != 0u) atomicAdd(o.counters_now + kDeepCounter, n);
  }
  if (t <= kListClasses) {  // thread c queues this workgroup's tiles of class c (kListClasses: the deep tiles): one atomic for all of them
    uint32_t n = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) n += (s_lit[w] != 0xFFFFFFFFu && (s_class[w] & 0xFFu) == t) ? 1u : 0u;
    if (n != 0u) {
      uint32_t slot = atomicAdd(o.counters_now + t, n);
      uint32_t* queue = o.lit_queue + (size_t)t * n_tiles;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (s_lit[w] != 0xFFFFFFFFu && (s_class[w] & 0xFFu) == t) queue[slot++] = s_lit[w];
    }
  }
#if defined(VKR_X_TIMES) && VKR_X_TIMES != 2  // (2: the array holds the deep-tile workgroups' stamps instead)
  if (lane == 0u && tile < n_tiles && tile < 32768u) {
    VKR_STAMP(7);  // queue push
#pragma unroll
    for (int i = 0; i < 8; ++i) g_wave_times[8u * tile + i] = stamp[i];
  }
#endif
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
// positions it could light (position_error_bound; it grows with the square of the distance from the camera).
// Measured at 8K x 4096 lights (2040 bins): 16 us, of which 4 us are launch and the plane set-up and 6 us the wait
// for the scattered index stores at the end.  Requesting the positions of eight trips together (20 us), and a 2 x 2
// group of bins per workgroup with the plane coefficients in scalar registers (20 us: 510 workgroups leave each
// SIMD two waves, and a lone wave issues one instruction per ~5 clocks), were both slower.
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
  const float* M = p.inv_vp;
  const float terms_w = (fabsf(M[3]) + fabsf(M[7])) + (fabsf(M[11]) + fabsf(M[15]));  // |cx|, |cy|, |depth| <= 1
  const float terms_xyz = fmaxf(fmaxf((fabsf(M[0]) + fabsf(M[4])) + (fabsf(M[8]) + fabsf(M[12])), (fabsf(M[1]) + fabsf(M[5])) + (fabsf(M[9]) + fabsf(M[13]))),
                                (fabsf(M[2]) + fabsf(M[6])) + (fabsf(M[10]) + fabsf(M[14])));
  uint32_t count = 0;  // block-uniform
  for (uint32_t base = 0; base < L; base += 256u) {
    const uint32_t i = base + t;
    bool hit = i < L;
    if (hit) {
      const float4 lp = lights[3u * i];
      // error of the positions the light could reach (position_error_bound): their coordinates are within the radius
      // of the light's, and 1 / |w| of a pixel's clip-space point is its view depth, at most its distance from the
      // camera (inverseVP being the inverse of a view-projection centred on cameraPos, as deferred.comp:118 takes it)
      const float coord = fmaxf(fmaxf(fabsf(lp.x), fabsf(lp.y)), fabsf(lp.z)) + radius;
      const float ex = lp.x - p.cam[0], ey = lp.y - p.cam[1], ez = lp.z - p.cam[2];
      const float dist = sqrt_(fma_(ez, ez, fma_(ey, ey, ex * ex))) + radius;
      const float reach = radius + (kPositionEps * fma_(fma_(coord, terms_w, terms_xyz), dist, coord) + 1e-3f);
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

// Below this many lights every tile tests the whole table.  Measured at 4K, radius 25, with / without the coarse
// bins: 256 lights 38.5 / 38.2 us, 512: 42.2 / 46.3, 1024: 50.4 / 64.8, 2048: 62.7 / 92.0, 4096: 85.9 / 156.5.
constexpr uint32_t kBinMinLights = 256;
constexpr uint64_t kBinMaxEntries = 64ull << 20;  // 256 MiB of indices
constexpr uint32_t kListCap = kChunk;             // entries per tile list: one LDS chunk
constexpr uint32_t kListMaxLights = 65536;        // list entries are 16-bit light indices
#ifndef VKR_X_FINE_ROUNDS
#define VKR_X_FINE_ROUNDS 4
#endif
constexpr uint32_t kFineRounds = VKR_X_FINE_ROUNDS;               // whole-table launches of fewer rounds of four-wave workgroups than this shade one pixel per lane

#ifndef VKR_X_WALK_SMALL_TILES
#define VKR_X_WALK_SMALL_TILES 0
#endif
// Below this many lights the loops over every light are the faster ones (the lists are built 32 lights at a time and
// every pixel pays for its sort).  Measured, lists / loops: 4K x 13 lights 0.212 / 0.191 ms, x 16 0.213 / 0.201, x 20
// 0.238 / 0.228, x 24 0.259 / 0.260, x 32 0.293 / 0.327; 1080p x 16 0.064 / 0.059, x 24 0.072 / 0.078, x 32 0.080 / 0.095.
#ifndef VKR_X_WALK_MIN
#define VKR_X_WALK_MIN 24
#endif
constexpr uint32_t kWalkMinLights = VKR_X_WALK_MIN;

template <int kOutMode>
hipError_t launch_mode(const DeferredParams& p, const TileListOut* lists, uint32_t cu_count, bool deep_tiles_expected, hipStream_t stream) {
  const dim3 grid((p.width + 31u) / 32u, (p.height + 31u) / 32u, 1u);
  const bool radius = (p.flags & 1u) != 0u;
  // fewer than kFineRounds rounds of four-wave workgroups (four of them fit a CU): one pixel per lane, single-wave workgroups
  const bool fine = (p.flags & 4u) == 0u && (uint64_t)grid.x * grid.y < (uint64_t)kFineRounds * 4u * cu_count;
  const dim3 fine_grid((p.width + 63u) / 64u, p.height, 1u);  // one wave per 64 pixels of a row
  const bool walk = !radius && (p.flags & 4u) == 0u && p.light_count >= kWalkMinLights;
  if (lists == nullptr && walk) {
    // Which tile.  A launch takes about (workgroups / workgroups resident + tail) workgroup lifetimes: the small tile's
    // workgroup (four resident per CU) lives 1.1 times as long as the large one's (two per CU; half the pixels, but
    // fewer pixels to sort make longer trips) and leaves a tail of 0.35 lifetimes against 0.5.  Fitted to 4K x 256, 270 / 540 / 2160
    // rows: small tiles 0.207 / 0.382 / 1.42 ms, large tiles 0.217 / 0.356 / 1.345 ms.  A short launch (a screen band)
    // and a launch whose last row of large tiles would be mostly empty take the small tile.
    const bool chunked = p.light_count > kWalkLights;
    const dim3 small_grid(grid.x, (p.height + 15u) / 16u, 1u);
    const float cus = (float)max(cu_count, 1u);
    const float small_cost = 1.1f * ((float)((uint64_t)small_grid.x * small_grid.y) / (4.0f * cus) + 0.35f);
    const float large_cost = (float)((uint64_t)grid.x * grid.y) / (2.0f * cus) + 0.5f;
    if (small_cost < large_cost || VKR_X_WALK_SMALL_TILES) {
      if (chunked)
        hipLaunchKernelGGL((deferred_walk_kernel<kOutMode, 16, true>), small_grid, dim3(256), 0, stream, p);
      else
        hipLaunchKernelGGL((deferred_walk_kernel<kOutMode, 16, false>), small_grid, dim3(256), 0, stream, p);
    } else {
      if (chunked)
        hipLaunchKernelGGL((deferred_walk_kernel<kOutMode, 32, true>), grid, dim3(512), 0, stream, p);
      else
        hipLaunchKernelGGL((deferred_walk_kernel<kOutMode, 32, false>), grid, dim3(512), 0, stream, p);
    }
  } else if (lists == nullptr) {
    if (!radius && !fine) hipLaunchKernelGGL((deferred_kernel<kOutMode, false, 4>), grid, dim3(256), 0, stream, p);
    if (!radius && fine) hipLaunchKernelGGL((deferred_pixel_kernel<kOutMode, false>), fine_grid, dim3(64), 0, stream, p);
    if (radius && !fine) hipLaunchKernelGGL((deferred_kernel<kOutMode, true, 4>), grid, dim3(256), 0, stream, p);
    if (radius && fine) hipLaunchKernelGGL((deferred_pixel_kernel<kOutMode, true>), fine_grid, dim3(64), 0, stream, p);
  } else {
    const uint32_t tiles = p.tiles_x * p.tiles_y;
    hipLaunchKernelGGL((tile_lists_kernel<kOutMode>), dim3((tiles + 3u) / 4u), dim3(256), 0, stream, p, *lists);
    if (hipError_t e = hipGetLastError()) return e;
    // The list launch is bound by HBM and the shading launch by the vector units, and they run one after the other.
    // Splitting the frame in two and putting the second half's two launches on a second stream, behind an event
    // after the first half's lists, so that they could run beside the first half's shading, was measured: 62-79 /
    // 110 / 192 us against 38.5 / 86 / 169 us (4K x 256, 4K x 4096, 8K x 4096) - the shading waves hold every
    // vector register of a SIMD, nothing of the other launch fits beside them, and the events cost their own time.
    // Strips: four grids' worth of waves (a CU holds four workgroups of four): the hardware starts the later ones as
    // the earlier ones finish, which evens out what the static deal leaves uneven (measured 8K x 4096 analytic /
    // uncorrelated depth: x1 0.198 / 1.41-1.78 ms, x2 0.189 / 1.36, x4 0.190 / 1.31, x8 0.191 / 1.29, x64 0.231 / 1.34).
    const uint32_t strip_groups = min((kStripsPerTile * tiles + 3u) / 4u, cu_count * 16u);
    const uint32_t deep_groups = deep_tiles_expected ? min(tiles, cu_count * 4u) : 0u;  // one workgroup per deep tile; a CU holds four
    hipLaunchKernelGGL((deferred_lists_kernel<kOutMode>), dim3(deep_groups + strip_groups), dim3(256), 0, stream, p, deep_groups);
  }
  return hipGetLastError();
}

}  // namespace

#ifdef VKR_X_TIMES
extern "C" __attribute__((visibility("default"))) int vkr_x_wave_times(unsigned long long* out, unsigned waves) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_times), (size_t)waves * 32u) == hipSuccess ? 0 : 1;
}
#endif
#ifdef VKR_X_COUNT
extern "C" __attribute__((visibility("default"))) int vkr_x_trip_counts(unsigned long long out[3], int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trip_counts), sizeof(g_trip_counts)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long zero[3] = {0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_trip_counts), zero, sizeof(zero)) != hipSuccess) return 1;
  }
  return 0;
}
#endif

hipError_t vkr_launch_exact_div_check(uint32_t n, uint32_t seed, uint32_t* mismatches, hipStream_t stream) {
  hipLaunchKernelGGL(exact_div_check_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, n, seed, mismatches);
  return hipGetLastError();
}

TiledLayout vkr_deferred_tiled_layout(const DeferredParams& p) {
  TiledLayout t{};
  const bool tiled = (p.flags & 1u) != 0u && (p.flags & 2u) == 0u && p.light_count <= kListMaxLights && p.width != 0u && p.height != 0u;
  if (!tiled) return t;
  t.tiled = true;
  t.tiles_x = (p.width + 31u) / 32u;
  t.tiles_y = (p.height + 31u) / 32u;
  t.list_cap = kListCap;
  const uint64_t tiles = (uint64_t)t.tiles_x * t.tiles_y;
  auto up = [](uint64_t v) { return (v + 255u) / 256u * 256u; };
  uint64_t off = 0;
  if (p.light_count >= kBinMinLights) {
    t.bins_x = (p.width + kBinPixels - 1u) / kBinPixels;
    t.bins = (uint64_t)t.bins_x * ((p.height + kBinPixels - 1u) / kBinPixels);
    t.bin_entries = t.bins * p.light_count;
    if (t.bin_entries > kBinMaxEntries) t.bins = t.bin_entries = 0, t.bins_x = 0;
  }
  t.off_bin_list = off;
  off += up(t.bin_entries * 4u);
  t.off_bin_count = off;
  off += up(t.bins * 4u);
  t.off_tile_count = off;
  off += up(tiles * 4u);
  t.off_lit_queue = off;
  off += up(tiles * 4u * (kListClasses + 1u));
  t.off_tile_list = off;
  off += up(tiles * t.list_cap * 2u);
  t.off_tile_order = off;
  off += up(tiles * 1024u * 2u);
  t.off_group_list = off;
  off += up(tiles * 8u * kGroupListCap * 2u);
  t.off_group_count = off;
  off += up(tiles * 8u * 2u);
  t.total_bytes = off;
  return t;
}

hipError_t vkr_launch_deferred(const DeferredParams& params, void* tiled_ws, uint32_t* counters, uint32_t epoch, uint32_t cu_count,
                               bool deep_tiles_expected, uint32_t* deep_hint, hipStream_t stream) {
  DeferredParams p = params;
  p.deep_hint = deep_hint;
  p.flags = (p.flags & ~kFlagNoDeepTiles) | (deep_tiles_expected ? 0u : kFlagNoDeepTiles);
  p.bin_list = nullptr;
  p.bin_count = nullptr;
  p.bins_x = 0;
  p.tile_list = nullptr;
  p.tile_count = nullptr;
  p.lit_queue = nullptr;
  p.lit_counters = nullptr;
  p.tile_order = nullptr;
  p.group_list = nullptr;
  p.group_count = nullptr;
  p.tiles_x = (p.width + 31u) / 32u;
  p.tiles_y = (p.height + 31u) / 32u;
  p.list_cap = 0;
  const TiledLayout t = vkr_deferred_tiled_layout(p);
  TileListOut lists{};
  const bool use_lists = t.tiled && tiled_ws && counters;
  if (t.tiled && !use_lists) return hipErrorInvalidValue;  // the caller provides the workspace of a tiled launch
  if (use_lists) {
    uint8_t* ws = static_cast<uint8_t*>(tiled_ws);
    if (t.bin_entries != 0u) {
      uint32_t* bin_list = reinterpret_cast<uint32_t*>(ws + t.off_bin_list);
      uint32_t* bin_count = reinterpret_cast<uint32_t*>(ws + t.off_bin_count);
      const dim3 bin_grid(t.bins_x, (uint32_t)(t.bins / t.bins_x), 1u);
      hipLaunchKernelGGL(light_bins_kernel, bin_grid, dim3(256), 0, stream, p, bin_list, bin_count);
      if (hipError_t e = hipGetLastError()) return e;
      p.bin_list = bin_list;
      p.bin_count = bin_count;
      p.bins_x = t.bins_x;
    }
    lists.tile_list = reinterpret_cast<uint16_t*>(ws + t.off_tile_list);
    lists.tile_count = reinterpret_cast<uint32_t*>(ws + t.off_tile_count);
    lists.lit_queue = reinterpret_cast<uint32_t*>(ws + t.off_lit_queue);
    lists.counters_now = counters + (epoch & 1u) * kListCounterStride;
    lists.counters_next = counters + ((epoch + 1u) & 1u) * kListCounterStride;
    lists.tile_order = reinterpret_cast<uint16_t*>(ws + t.off_tile_order);
    p.tile_order = lists.tile_order;
    lists.group_list = reinterpret_cast<uint16_t*>(ws + t.off_group_list);
    lists.group_count = reinterpret_cast<uint16_t*>(ws + t.off_group_count);
    p.group_list = lists.group_list;
    p.group_count = lists.group_count;
    p.tile_list = lists.tile_list;
    p.tile_count = lists.tile_count;
    p.lit_queue = lists.lit_queue;
    p.lit_counters = lists.counters_now;
    p.list_cap = t.list_cap;
  }
  switch (p.out_mode) {
    case kOutRgba8: return launch_mode<kOutRgba8>(p, use_lists ? &lists : nullptr, cu_count, deep_tiles_expected, stream);
    case kOutBgra8: return launch_mode<kOutBgra8>(p, use_lists ? &lists : nullptr, cu_count, deep_tiles_expected, stream);
    case kOutRgba16f: return launch_mode<kOutRgba16f>(p, use_lists ? &lists : nullptr, cu_count, deep_tiles_expected, stream);
    case kOutRgba32f: return launch_mode<kOutRgba32f>(p, use_lists ? &lists : nullptr, cu_count, deep_tiles_expected, stream);
    default: return hipErrorInvalidValue;
  }
}
