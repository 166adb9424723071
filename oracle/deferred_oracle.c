/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement ("oracle") of the deferred lighting
 * resolve.  Nothing under oracle/ is part of the shipped path; only tests, the
 * smoke check and the benchmark's CPU-baseline leg may load this library.
 *
 * PINNED AGAINST THE SHADER'S TEXT, NOT AGAINST A VULKAN DRIVER: the reference
 * ships no tests, golden vectors or fixtures for this pass, and its shader
 * cannot be executed by a driver in this environment (no Vulkan loader/ICD, no
 * GLSL compiler).  This file is a statement-by-statement restatement of
 * /root/reference/shaders/deferred.comp:1-131.  What pins it: the reference's
 * own text, evaluated per pixel in binary64 by tests/glsl_eval.py (a small
 * GLSL-subset evaluator; tests/test_shader_text_cpu.py and the fixture
 * tests/golden/shader_text_eval_*.npz made from such a run) - oracle_deferred_f64
 * agrees with it to 1e-12, and oracle_deferred_f32 is bit for bit the text
 * evaluated with every operation rounded to binary32 - besides the known-answer
 * cases in tests/.  How a driver rounds pow / normalize / division remains
 * outside any pin.
 *
 * Three evaluations of the same algorithm are built from one body
 * (deferred_oracle_body.inc):
 *
 *   oracle_deferred_f32   every operation in IEEE binary32, in GLSL evaluation
 *                         order (left-to-right association, no FMA contraction:
 *                         compile with -ffp-contract=off, no -ffast-math).
 *   oracle_deferred_f64   every operation in binary64 (error-budget studies).
 *   oracle_deferred_mix   the depth->world-position step (deferred.comp:99-103)
 *                         in binary32 exactly as oracle_deferred_f32 does it,
 *                         everything after it in binary64.  Position
 *                         reconstruction from a [0,1] depth through an fp32
 *                         inverse view-projection is ill-conditioned (w = a*z+b
 *                         cancels for z -> 1), so the fp32 positions are taken
 *                         as the reference's inputs to lighting, and the
 *                         lighting itself is evaluated (nearly) exactly.
 *
 * Conventions fixed here (where GLSL/SPIR-V leave latitude):
 *   dot(a,b)        = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   mat4*vec4 row r = ((m[0][r]*x + m[1][r]*y) + m[2][r]*z) + m[3][r]*w
 *   normalize(x)    = x * (1/sqrt(dot(x,x)))
 *   mix(x,y,a)      = x*(1-a) + y*a
 *   pow(x,5)        = pow{f,}(x, 5)
 *   rgba8 store     = round-half-even(clamp(c,0,1)*255), NaN -> 0, alpha 255
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* Constants exactly as compiled into the reference SPIR-V (deferred.comp:2-3,
 * :43, :106, :116, :122 and the folded 1.0/1.51 of :49). */
#define K_PI 3.14159f
#define K_AMBIENT 0.05f
#define K_ALPHA_FLOOR 0.0064f
#define K_EARLY_OUT 1e-6f
#define K_DIST_FLOOR 1e-7f

static float k_inv151(void) {
  /* 1.0/1.51 folded by glslang to binary32 0x3F298953. */
  uint32_t bits = 0x3F298953u;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static float half_to_float(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {
      /* subnormal half: normalise */
      int e = -1;
      do {
        man <<= 1;
        ++e;
      } while ((man & 0x400u) == 0);
      man &= 0x3FFu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

static uint8_t quantize_unorm8(double c) {
  /* Vulkan float -> unorm8: NaN -> 0, clamp to [0,1], scale, round to nearest
   * even.  (deferred.comp:23,130: rgba8 image store.) */
  if (!(c > 0.0)) return 0;
  if (c > 1.0) c = 1.0;
  return (uint8_t)nearbyint(c * 255.0);
}

static uint8_t quantize_unorm8_f32(float c) {
  if (!(c > 0.0f)) return 0;
  if (c > 1.0f) c = 1.0f;
  return (uint8_t)nearbyintf(c * 255.0f);
}

/* Arguments shared by the three evaluations.  Planes are linear, row-major;
 * pitches are in bytes.  `lights` uses the reference's 48-byte std140 record
 * (deferred.comp:12-19): pos @0, color @16; the other fields are ignored, as
 * the shader ignores them.  `push` is the 80-byte block of deferred.comp:6-10:
 * cameraPos @0, pointLightRadius @12, inverseVP @16 (column-major).
 *
 * rows [y0,y1) of a W x H frame are evaluated; out pointers address row 0 of
 * the FULL frame (only rows y0..y1-1 are written).
 *
 * flags bit 0: radius cutoff extension (not in the reference): a light
 * contributes only if dot(Lv,Lv) < pointLightRadius^2, tested in binary32.
 *
 * near_threshold (optional, W*H bytes): set to 1 where dot(viewDir,normal) is
 * within 5e-7 of the 1e-6 early-out threshold (the specular term is
 * discontinuous there, so implementations may legitimately differ). */
typedef struct {
  const uint8_t* albedo;
  const uint8_t* material;
  const uint16_t* normal;
  const float* depth;
  int64_t pitch_albedo, pitch_material, pitch_normal, pitch_depth;
  int32_t width, height;
  const float* lights;
  int32_t light_count;
  const float* push;
  int32_t y0, y1;
  uint32_t flags;
  float* out_f32;      /* W*H*4 floats (rgba, a = 1) or NULL */
  double* out_f64;     /* W*H*4 doubles or NULL */
  uint8_t* out_rgba8;  /* W*H*4 bytes or NULL */
  uint8_t* near_threshold;
  float* view_sensitivity; /* W*H*3 floats or NULL: see the conditioning probe in deferred_oracle_body.inc */
} oracle_deferred_args;

#define REAL float
#define POSREAL float
#define FN(name) name##_f32
#define ORACLE_NAME oracle_deferred_f32
#define R_SQRT sqrtf
#define R_POW powf
#define R_MAX fmaxf
#define R_MIN fminf
#include "deferred_oracle_body.inc"
#undef REAL
#undef POSREAL
#undef FN
#undef ORACLE_NAME
#undef R_SQRT
#undef R_POW
#undef R_MAX
#undef R_MIN

#define REAL double
#define POSREAL double
#define FN(name) name##_f64
#define ORACLE_NAME oracle_deferred_f64
#define R_SQRT sqrt
#define R_POW pow
#define R_MAX fmax
#define R_MIN fmin
#include "deferred_oracle_body.inc"
#undef REAL
#undef POSREAL
#undef FN
#undef ORACLE_NAME
#undef R_SQRT
#undef R_POW
#undef R_MAX
#undef R_MIN

#define REAL double
#define POSREAL float
#define FN(name) name##_mix
#define ORACLE_NAME oracle_deferred_mix
#define R_SQRT sqrt
#define R_POW pow
#define R_MAX fmax
#define R_MIN fmin
#include "deferred_oracle_body.inc"
