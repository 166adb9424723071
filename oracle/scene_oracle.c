/* CPU restatement of the callers either side of the two passes -- TEST INFRASTRUCTURE ONLY
 * (used by tests/ and by the checker legs of __graft_entry__.smoke() / bench.py; never linked
 * into the product).
 *
 * PARITY UNPINNED: the reference ships no tests or fixtures for any of this, and its host code
 * cannot be built here (assimp, glm, GLFW, nlohmann::json, vulkan.hpp are all absent).  Each
 * function follows the reference lines it cites; where the reference calls glm the published
 * formula is restated (glm is an empty, unpinned submodule).
 *
 *   scene tables      /root/reference/src/vkmodel.cpp:55-111
 *   mesh / world AABB /root/reference/src/vkmodel.cpp:166-174 (+ cull_oracle.c)
 *   camera            /root/reference/inc/vkcamera.hpp:11-56,59-65
 *   frame constants   /root/reference/src/main.cpp:206-209
 *   light records     /root/reference/src/main.cpp:40-47,57-69
 *   attachment writes /root/reference/shaders/offscreen.frag:13-16, src/main.cpp:112-116,125-129
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void oracle_world_aabb(const float* corners24, const float* model, float out_mn[3], float out_mx[3]);
void oracle_compute_aabb_corners(const float* positions, uint64_t count, float* corners24);

/* column-major 4x4 product in glm's association: ((a0*b0 + a1*b1) + a2*b2) + a3*b3 */
static void mat_mul(const float* a, const float* b, float* out) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int row = 0; row < 4; ++row) {
      float acc = 0.0f;
      for (int k = 0; k < 4; ++k) acc += a[k * 4 + row] * b[c * 4 + k];
      r[c * 4 + row] = acc;
    }
  memcpy(out, r, sizeof(r));
}

/* ---------------------------------------------------------------- scene tables */

typedef struct {
  uint32_t node_count;
  const float* node_transform;      /* a1..d4 per node */
  const uint32_t* node_child_begin; /* CSR */
  const uint32_t* node_children;
  const uint32_t* node_mesh_begin;
  const uint32_t* node_meshes;
  uint32_t mesh_count;
  const uint32_t* mesh_vertex_count;
  const uint32_t* mesh_face_count;
  const uint32_t* mesh_material;
  const uint32_t* mesh_valid; /* 1 = HasFaces && HasTextureCoords(0) && triangles (vkmodel.cpp:91) */
} oracle_scene;

/* vkmodel.cpp:55-73.  Emits (mesh, world matrix) pairs in visiting order into `pairs_mesh` /
 * `pairs_matrix` (capacity = number of node-mesh references); returns their count. */
uint32_t oracle_scene_flatten(const oracle_scene* s, uint32_t* pairs_mesh, float* pairs_matrix) {
  typedef struct {
    uint32_t node;
    float parent[16];
  } item;
  item* stack = (item*)malloc(sizeof(item) * (s->node_count + 1));
  uint32_t top = 0, n = 0;
  stack[top].node = 0;
  for (int i = 0; i < 16; ++i) stack[top].parent[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  ++top;
  while (top) {
    item it = stack[--top];
    const float* t = s->node_transform + 16 * (size_t)it.node;
    /* glm::mat4{a1,a2,a3,a4,b1,...} takes the sixteen values as columns; transpose() (:65) makes
     * (a1..a4) the first ROW, i.e. column-major element [c][r] = t[r*4+c] */
    float local[16], cur[16];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) local[c * 4 + r] = t[r * 4 + c];
    mat_mul(it.parent, local, cur); /* :66 */
    for (uint32_t k = s->node_mesh_begin[it.node]; k < s->node_mesh_begin[it.node + 1]; ++k) { /* :69-70 */
      pairs_mesh[n] = s->node_meshes[k];
      memcpy(pairs_matrix + 16 * (size_t)n, cur, sizeof(cur));
      ++n;
    }
    for (uint32_t k = s->node_child_begin[it.node]; k < s->node_child_begin[it.node + 1]; ++k) { /* :71-72 */
      stack[top].node = s->node_children[k];
      memcpy(stack[top].parent, cur, sizeof(cur));
      ++top;
    }
  }
  free(stack);
  return n;
}

/* vkmodel.cpp:89-111 given the flattened pairs.  Output arrays sized for mesh_count (+1) meshes
 * and pair_count matrices; returns the number of kept meshes.  totals = {vertices, indices,
 * matrices}. */
uint32_t oracle_scene_tables(const oracle_scene* s, const uint32_t* pairs_mesh, const float* pairs_matrix, uint32_t pair_count,
                             uint32_t* mesh_info /*4 per*/, uint32_t* draw_cmd /*5 per*/, uint32_t* per_mesh /*2 per*/,
                             float* matrices, uint32_t* mesh_base, uint32_t* vertex_base, uint32_t* source_mesh, uint32_t* totals) {
  uint32_t kept = 0, tv = 0, ti = 0, tm = 0;
  for (uint32_t m = 0; m < s->mesh_count; ++m) {
    if (!s->mesh_valid[m]) continue; /* :91-93 */
    const uint32_t index_count = 3u * s->mesh_face_count[m];
    uint32_t inst = 0;
    for (uint32_t p = 0; p < pair_count; ++p) /* instanceTransforms[meshIdx], in push_back order */
      if (pairs_mesh[p] == m) {
        memcpy(matrices + 16 * (size_t)(tm + inst), pairs_matrix + 16 * (size_t)p, 64);
        ++inst;
      }
    mesh_info[4 * kept + 0] = index_count; /* :95 */
    mesh_info[4 * kept + 1] = ti;
    mesh_info[4 * kept + 2] = tv;
    mesh_info[4 * kept + 3] = s->mesh_material[m];
    draw_cmd[5 * kept + 0] = index_count; /* :96 */
    draw_cmd[5 * kept + 1] = inst;
    draw_cmd[5 * kept + 2] = ti;
    draw_cmd[5 * kept + 3] = tv;
    draw_cmd[5 * kept + 4] = 0;
    per_mesh[2 * kept + 0] = s->mesh_material[m]; /* :108 */
    per_mesh[2 * kept + 1] = tm;
    mesh_base[kept] = tm;
    vertex_base[kept] = tv;
    source_mesh[kept] = m;
    tv += s->mesh_vertex_count[m]; /* :97-98 */
    ti += index_count;
    tm += inst;
    ++kept;
  }
  mesh_base[kept] = tm;
  vertex_base[kept] = tv;
  totals[0] = tv;
  totals[1] = ti;
  totals[2] = tm;
  return kept;
}

/* Per-mesh min/max (computeAABB, :166-171) and per-instance world bounds (corners :172-173 under
 * the model matrix).  mesh_aabb: 8 floats per mesh (min.xyz,0,max.xyz,0); aabb_min/max: 4 floats
 * per instance, aabb_min.w = mesh id bits. */
void oracle_mesh_aabbs(const float* positions, const uint32_t* vertex_base, uint32_t mesh_count, float* mesh_aabb) {
  for (uint32_t m = 0; m < mesh_count; ++m) {
    float* o = mesh_aabb + 8 * (size_t)m;
    const uint32_t n = vertex_base[m + 1] - vertex_base[m];
    if (n == 0) {
      o[0] = o[1] = o[2] = INFINITY;
      o[4] = o[5] = o[6] = -INFINITY;
    } else {
      float corners[24];
      oracle_compute_aabb_corners(positions + 3 * (size_t)vertex_base[m], n, corners);
      memcpy(o, corners, 12);          /* corner 0 = (mn,mn,mn) */
      memcpy(o + 4, corners + 21, 12); /* corner 7 = (mx,mx,mx) */
    }
    o[3] = 0.0f;
    o[7] = 0.0f;
  }
}

void oracle_instance_aabbs(const float* matrices, const uint32_t* mesh_base, const float* mesh_aabb, uint32_t mesh_count,
                           float* aabb_min, float* aabb_max) {
  for (uint32_t m = 0; m < mesh_count; ++m) {
    const float* lo = mesh_aabb + 8 * (size_t)m;
    const float* hi = lo + 4;
    const float* sel[2] = {lo, hi};
    static const int pick[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
    float corners[24];
    for (int k = 0; k < 8; ++k)
      for (int c = 0; c < 3; ++c) corners[3 * k + c] = sel[pick[k][c]][c];
    for (uint32_t i = mesh_base[m]; i < mesh_base[m + 1]; ++i) {
      float mn[3], mx[3];
      oracle_world_aabb(corners, matrices + 16 * (size_t)i, mn, mx);
      memcpy(aabb_min + 4 * (size_t)i, mn, 12);
      memcpy(aabb_max + 4 * (size_t)i, mx, 12);
      memcpy(aabb_min + 4 * (size_t)i + 3, &m, 4);
      aabb_max[4 * (size_t)i + 3] = 0.0f;
    }
  }
}

/* ---------------------------------------------------------------- camera (vkcamera.hpp) */

typedef struct {
  float pos[3], forward[3], up[3];
} oracle_camera;

static void cross3(const float* a, const float* b, float* o) {
  const float x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x;
  o[1] = y;
  o[2] = z;
}
static float dot3(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

void oracle_camera_default(oracle_camera* c) { /* :62-63 */
  const oracle_camera d = {{0.0f, 3.0f, 71.0f}, {0.0f, 0.0f, -1.0f}, {0.0f, -1.0f, 0.0f}};
  *c = d;
}

/* glm::lookAt (right-handed) */
void oracle_look_at(const float* eye, const float* center, const float* up, float* m) {
  float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
  const float fl = 1.0f / sqrtf(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
  for (int i = 0; i < 3; ++i) f[i] *= fl;
  float s[3];
  cross3(f, up, s);
  const float sl = 1.0f / sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
  for (int i = 0; i < 3; ++i) s[i] *= sl;
  float u[3];
  cross3(s, f, u);
  m[0] = s[0]; m[4] = s[1]; m[8] = s[2];
  m[1] = u[0]; m[5] = u[1]; m[9] = u[2];
  m[2] = -f[0]; m[6] = -f[1]; m[10] = -f[2];
  m[3] = m[7] = m[11] = 0.0f;
  m[12] = -(s[0] * eye[0] + s[1] * eye[1] + s[2] * eye[2]);
  m[13] = -(u[0] * eye[0] + u[1] * eye[1] + u[2] * eye[2]);
  m[14] = f[0] * eye[0] + f[1] * eye[1] + f[2] * eye[2];
  m[15] = 1.0f;
}

void oracle_camera_view(const oracle_camera* c, float* view) { /* :51-53 */
  const float center[3] = {c->pos[0] + c->forward[0], c->pos[1] + c->forward[1], c->pos[2] + c->forward[2]};
  oracle_look_at(c->pos, center, c->up, view);
}

/* :30-47; key: 0 W, 1 A, 2 S, 3 D, 4 SPACE, 5 LEFT_SHIFT */
void oracle_camera_key(oracle_camera* c, int key, float delta_time) {
  if (key < 0 || key > 5) return;
  float mult = 0.2f * delta_time;
  if (key == 4 || key == 1 || key == 2) mult *= -1.0f;
  float dir[3];
  cross3(c->forward, c->up, dir);
  if (key == 5 || key == 4) memcpy(dir, c->up, 12);
  else if (key == 0 || key == 2) memcpy(dir, c->forward, 12);
  for (int i = 0; i < 3; ++i) c->pos[i] += dir[i] * mult;
}

/* :11-27 for a cursor delta (dx, dy); a zero delta is left as a no-op */
void oracle_camera_cursor_delta(oracle_camera* c, float dx, float dy) {
  const float deg60 = 60.0f * 0.01745329251994329576923690768489f;
  const float pi = 3.14159265358979323846f;
  const float cur = acosf(dot3(c->up, c->forward));
  float nu = cur - dy * deg60;
  const float lo = 0.01f * pi, hi = 0.99f * pi;
  nu = nu < lo ? lo : (nu > hi ? hi : nu);
  dy = (cur - nu) / deg60;
  const float len = sqrtf(dx * dx + dy * dy);
  if (!(len > 0.0f)) return;
  float r[3], u[3], axis[3];
  cross3(c->forward, c->up, r);
  cross3(r, c->forward, u);
  for (int i = 0; i < 3; ++i) axis[i] = r[i] * dx + u[i] * dy;
  cross3(c->forward, axis, axis);
  const float angle = deg60 * len;
  /* glm::rotate(identity, angle, axis) */
  const float co = cosf(angle), si = sinf(angle);
  const float inv = 1.0f / sqrtf(dot3(axis, axis));
  const float a[3] = {axis[0] * inv, axis[1] * inv, axis[2] * inv};
  const float t[3] = {(1.0f - co) * a[0], (1.0f - co) * a[1], (1.0f - co) * a[2]};
  const float R0[3] = {co + t[0] * a[0], t[0] * a[1] + si * a[2], t[0] * a[2] - si * a[1]};
  const float R1[3] = {t[1] * a[0] - si * a[2], co + t[1] * a[1], t[1] * a[2] + si * a[0]};
  const float R2[3] = {t[2] * a[0] + si * a[1], t[2] * a[1] - si * a[0], co + t[2] * a[2]};
  const float f[3] = {c->forward[0], c->forward[1], c->forward[2]};
  for (int i = 0; i < 3; ++i) c->forward[i] = (R0[i] * f[0] + R1[i] * f[1]) + R2[i] * f[2];
}

/* ---------------------------------------------------------------- lights (main.cpp:57-69) */
void oracle_pack_point_lights(uint32_t n, const float* pos, const float* intensity, const float* dir, const float* open_deg,
                              const float* pen_deg, float* out /* 12 floats per light */) {
  const float rad = 0.01745329251994329576923690768489f;
  for (uint32_t i = 0; i < n; ++i) {
    float* o = out + 12 * (size_t)i;
    memcpy(o + 0, pos + 3 * (size_t)i, 12);
    o[3] = cosf((open_deg[i] * rad) / 2.0f);
    memcpy(o + 4, intensity + 3 * (size_t)i, 12);
    o[7] = cosf((pen_deg[i] * rad) / 2.0f);
    memcpy(o + 8, dir + 3 * (size_t)i, 12);
    o[11] = 0.0f;
  }
}

/* ---------------------------------------------------------------- attachment write conversions */
static uint32_t unorm8(float c) { /* NaN -> 0, clamp, round half to even */
  const float t = (c > 0.0f) ? (c < 1.0f ? c : 1.0f) : 0.0f;
  return (uint32_t)lrintf(t * 255.0f); /* default rounding mode = to nearest even */
}

/* binary32 -> binary16, round to nearest even, overflow -> inf, subnormals kept */
static uint16_t half_bits(float f) {
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t a = x & 0x7FFFFFFFu;
  if (a >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((a > 0x7F800000u) ? (0x0200u | ((a >> 13) & 0x03FFu)) : 0u)); /* inf / NaN */
  if (a >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* rounds to >= 2^16 */
  if (a >= 0x38800000u) {                                  /* normal half */
    const uint32_t mant = a - 0x38000000u;                 /* rebias exponent */
    const uint32_t r = mant + 0x00000FFFu + ((mant >> 13) & 1u);
    return (uint16_t)(sign | (r >> 13));
  }
  if (a < 0x33000000u) return (uint16_t)sign; /* below half of the smallest subnormal (2^-25): ties go to even = 0 */
  /* subnormal half: value = m * 2^-24 */
  const int e = (int)(a >> 23);                 /* biased exponent, 102..112 */
  const uint32_t m = (a & 0x007FFFFFu) | 0x00800000u;
  const int shift = 126 - e;                    /* 14..24 */
  const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  const uint32_t r = q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u);
  return (uint16_t)(sign | r);
}

void oracle_gbuffer_encode(const float* albedo, const float* material, const float* normal, const float* depth, uint64_t pixels,
                           uint8_t* o_albedo, uint8_t* o_material, uint16_t* o_normal, float* o_depth) {
  for (uint64_t p = 0; p < pixels; ++p) {
    for (int c = 0; c < 4; ++c) {
      o_albedo[4 * p + c] = (uint8_t)unorm8(albedo[4 * p + c]);
      o_material[4 * p + c] = (uint8_t)unorm8(material[4 * p + c]);
      o_normal[4 * p + c] = half_bits(normal[4 * p + c]);
    }
    o_depth[p] = depth[p];
  }
}

/* offscreen.vert:28-35 + offscreen.frag:13-16 for one surface sample per pixel; see vkr_gbuffer_encode_surfaces for
 * the evaluation order.  normal_obj[4p+3] carries the matrix index as uint32 bits. */
static void rotate_dir3(const float* M, const float* v, float* o) {
  for (int r = 0; r < 3; ++r) o[r] = (M[0 + r] * v[0] + M[4 + r] * v[1]) + M[8 + r] * v[2];
}
static void normalize3(float* v) {
  const float len = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  v[0] = v[0] / len;
  v[1] = v[1] / len;
  v[2] = v[2] / len;
}
void oracle_gbuffer_encode_surfaces(const float* albedo_texel, const float* specular_texel, const float* normal_texel, const float* normal_obj,
                                    const float* tangent_obj, const float* depth, const float* matrices, uint32_t matrix_count, uint64_t pixels,
                                    uint8_t* o_albedo, uint8_t* o_material, uint16_t* o_normal, float* o_depth) {
  for (uint64_t p = 0; p < pixels; ++p) {
    uint32_t mi;
    memcpy(&mi, normal_obj + 4 * p + 3, 4);
    if (mi >= matrix_count) mi = 0;
    const float* M = matrices + 16 * (uint64_t)mi;
    float N[3], T[3], B[3], m[3], out[4];
    rotate_dir3(M, normal_obj + 4 * p, N);   /* offscreen.vert:32 */
    normalize3(N);
    rotate_dir3(M, tangent_obj + 4 * p, T);  /* :33 */
    normalize3(T);
    B[0] = N[1] * T[2] - N[2] * T[1];        /* :34 */
    B[1] = N[2] * T[0] - N[0] * T[2];
    B[2] = N[0] * T[1] - N[1] * T[0];
    for (int c = 0; c < 3; ++c) m[c] = normal_texel[4 * p + c] * 2.0f - 1.0f;  /* offscreen.frag:15 */
    normalize3(m);
    for (int c = 0; c < 3; ++c) out[c] = (T[c] * m[0] + B[c] * m[1]) + N[c] * m[2];
    out[3] = 1.0f;                           /* :16 */
    for (int c = 0; c < 4; ++c) {
      o_albedo[4 * p + c] = (uint8_t)unorm8(albedo_texel[4 * p + c]);     /* :13 */
      o_material[4 * p + c] = (uint8_t)unorm8(specular_texel[4 * p + c]); /* :14 */
      o_normal[4 * p + c] = half_bits(out[c]);
    }
    o_depth[p] = depth[p];
  }
}

/* ---------------------------------------------------------------- analytic scene fill
 * The build's own test-scene generator (no reference counterpart beyond the attachment formats and
 * clear values): spheres in table order, then the ground plane; nearest hit with 0 < t < 1 in units
 * of (far-plane point - camera). */
typedef struct {
  float center[3], radius, albedo[3], roughness, metalness, occlusion, pad[2];
} oracle_sphere;

typedef struct {
  float view_proj[16], inv_view_proj[16], camera_pos[3];
  uint32_t sphere_count;
  float ground_plane[4], ground_albedo_a[3], ground_tile, ground_albedo_b[3], ground_roughness, ground_metalness, pad[3];
} oracle_gbuffer_scene;

void oracle_gbuffer_fill(const oracle_gbuffer_scene* sc, const oracle_sphere* spheres, uint32_t width, uint32_t height, uint32_t row_offset,
                         uint32_t full_height, uint8_t* o_albedo, uint8_t* o_material, uint16_t* o_normal, float* o_depth) {
  const float* M = sc->inv_view_proj;
  const float* VP = sc->view_proj;
  const float* cam = sc->camera_pos;
  const float* gn = sc->ground_plane;
  const float gn2 = (gn[0] * gn[0] + gn[1] * gn[1]) + gn[2] * gn[2];
  for (uint32_t y = 0; y < height; ++y) {
    const float v = ((float)(y + row_offset) + 0.5f) / (float)full_height;
    const float cy = v * 2.0f - 1.0f;
    for (uint32_t x = 0; x < width; ++x) {
      const uint64_t p = (uint64_t)y * width + x;
      const float u = ((float)x + 0.5f) / (float)width;
      const float cx = u * 2.0f - 1.0f;
      const float wx = ((M[0] * cx + M[4] * cy) + M[8]) + M[12];
      const float wy = ((M[1] * cx + M[5] * cy) + M[9]) + M[13];
      const float wz = ((M[2] * cx + M[6] * cy) + M[10]) + M[14];
      const float ww = ((M[3] * cx + M[7] * cy) + M[11]) + M[15];
      const float d[3] = {wx / ww - cam[0], wy / ww - cam[1], wz / ww - cam[2]};
      float best = 1.0f;
      int64_t what = -1;
      const float a = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2];
      for (uint32_t s = 0; s < sc->sphere_count; ++s) {
        const float oc[3] = {cam[0] - spheres[s].center[0], cam[1] - spheres[s].center[1], cam[2] - spheres[s].center[2]};
        const float c = ((oc[0] * oc[0] + oc[1] * oc[1]) + oc[2] * oc[2]) - spheres[s].radius * spheres[s].radius;
        const float b = (oc[0] * d[0] + oc[1] * d[1]) + oc[2] * d[2];
        const float disc = b * b - a * c;
        if (disc >= 0.0f) {
          const float t = (-b - sqrtf(disc)) / a;
          if (t > 0.0f && t < best) {
            best = t;
            what = s;
          }
        }
      }
      if (gn2 > 0.0f) {
        const float dist = ((gn[0] * cam[0] + gn[1] * cam[1]) + gn[2] * cam[2]) + gn[3];
        const float denom = (gn[0] * d[0] + gn[1] * d[1]) + gn[2] * d[2];
        if (denom < 0.0f) {
          const float t = -dist / denom;
          if (t > 0.0f && t < best) {
            best = t;
            what = -2;
          }
        }
      }
      float alb[4], mat[4], nrm[4], dep;
      if (what == -1) { /* clear values, main.cpp:125-129 */
        alb[0] = alb[1] = alb[2] = 0.0f; alb[3] = 1.0f;
        mat[0] = mat[1] = mat[2] = 0.0f; mat[3] = 1.0f;
        nrm[0] = nrm[1] = nrm[2] = 0.0f; nrm[3] = 1.0f;
        dep = 1.0f;
      } else {
        const float P[3] = {cam[0] + d[0] * best, cam[1] + d[1] * best, cam[2] + d[2] * best};
        float N[3];
        if (what >= 0) {
          const oracle_sphere* s = &spheres[what];
          for (int k = 0; k < 3; ++k) N[k] = (P[k] - s->center[k]) / s->radius;
          alb[0] = s->albedo[0]; alb[1] = s->albedo[1]; alb[2] = s->albedo[2];
          mat[0] = s->occlusion; mat[1] = s->roughness; mat[2] = s->metalness;
        } else {
          const float inv = 1.0f / sqrtf(gn2);
          for (int k = 0; k < 3; ++k) N[k] = gn[k] * inv;
          const float q = floorf(P[0] / sc->ground_tile) + floorf(P[2] / sc->ground_tile);
          const int odd = (q - 2.0f * floorf(q * 0.5f)) != 0.0f;
          const float* g = odd ? sc->ground_albedo_b : sc->ground_albedo_a;
          alb[0] = g[0]; alb[1] = g[1]; alb[2] = g[2];
          mat[0] = 1.0f; mat[1] = sc->ground_roughness; mat[2] = sc->ground_metalness;
        }
        alb[3] = 1.0f;
        mat[3] = 1.0f;
        nrm[0] = -N[0]; nrm[1] = -N[1]; nrm[2] = -N[2]; nrm[3] = 1.0f;
        const float cz = ((VP[2] * P[0] + VP[6] * P[1]) + VP[10] * P[2]) + VP[14];
        const float cw = ((VP[3] * P[0] + VP[7] * P[1]) + VP[11] * P[2]) + VP[15];
        dep = cz / cw;
      }
      for (int c = 0; c < 4; ++c) {
        o_albedo[4 * p + c] = (uint8_t)unorm8(alb[c]);
        o_material[4 * p + c] = (uint8_t)unorm8(mat[c]);
        o_normal[4 * p + c] = half_bits(nrm[c]);
      }
      o_depth[p] = dep;
    }
  }
}
