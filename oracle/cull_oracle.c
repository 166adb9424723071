/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement ("oracle") of the per-instance
 * frustum cull + indirect-command compaction.  Only tests, the smoke check and
 * the benchmark's CPU-baseline leg may load this library.
 *
 * PARITY UNPINNED: /root/reference/shaders/prep_indirect.comp:1-5 is an empty
 * `void main() {}`; the reference has no cull arithmetic, output or fixture.
 * What the reference does fix is the data contract, which this file follows:
 *   - the 20-byte VkDrawIndexedIndirectCommand record, one per mesh, built at
 *     /root/reference/src/vkmodel.cpp:79,96 and consumed with stride 20 at
 *     /root/reference/inc/vkmodel.hpp:72;
 *   - instances of one mesh are contiguous in the model-matrix array, starting
 *     at PerMeshData::modelMatrixBaseIndex
 *     (/root/reference/inc/vkmodel.hpp:76-78, src/vkmodel.cpp:108-110,
 *     shaders/offscreen.vert:28-30);
 *   - clip position = VP * model * vec4(p,1) (shaders/offscreen.vert:31), with
 *     Vulkan clip rules -w<=x,y<=w, 0<=z<=w.
 *
 * The cull rule itself is defined by this build:
 *   instance i has a world-space AABB (mn, mx) and a mesh id; plane k is a
 *   float4 (nx,ny,nz,nw); p-vertex: px = nx >= 0 ? mx.x : mn.x (same for y,z);
 *   d = ((nx*px + ny*py) + nz*pz) + nw in binary32 with NO fused multiply-add;
 *   the instance is visible iff d >= 0 for all six planes (NaN => culled).
 *   Output: for mesh m with segment base b_m, the visible instance indices in
 *   ascending order at visible[b_m ...], and command m = template m with
 *   instanceCount = visible count and firstInstance = b_m.
 *
 * Compile with -ffp-contract=off and without -ffast-math.
 */
#include <stdint.h>
#include <string.h>

typedef struct {
  uint32_t indexCount;
  uint32_t instanceCount;
  uint32_t firstIndex;
  int32_t vertexOffset;
  uint32_t firstInstance;
} oracle_draw_cmd; /* VkDrawIndexedIndirectCommand, 20 bytes */

typedef struct {
  const float* aabb_min; /* N x float4: xyz = min corner, w = mesh id (uint32 bits) */
  const float* aabb_max; /* N x float4: xyz = max corner, w = unused */
  uint64_t instance_count;
  const float* planes; /* 6 x float4 */
  const uint32_t* mesh_base; /* M: first instance of each mesh */
  uint32_t mesh_count;
  const oracle_draw_cmd* cmd_template; /* M */
  oracle_draw_cmd* cmd_out;            /* M */
  uint32_t* visible;                   /* N: compacted per mesh segment; untouched slots keep their value */
  uint8_t* mask;                       /* optional, N bytes: 1 = visible */
} oracle_cull_args;

int oracle_aabb_visible(const float mn[3], const float mx[3], const float* planes) {
  for (int k = 0; k < 6; ++k) {
    const float nx = planes[4 * k + 0], ny = planes[4 * k + 1], nz = planes[4 * k + 2], nw = planes[4 * k + 3];
    const float px = nx >= 0.0f ? mx[0] : mn[0];
    const float py = ny >= 0.0f ? mx[1] : mn[1];
    const float pz = nz >= 0.0f ? mx[2] : mn[2];
    const float t0 = nx * px;
    const float t1 = ny * py;
    const float t2 = nz * pz;
    const float d = ((t0 + t1) + t2) + nw;
    if (!(d >= 0.0f)) return 0;
  }
  return 1;
}

/* returns 0 on success, 1 if an instance carries a mesh id >= mesh_count */
int oracle_cull(const oracle_cull_args* a) {
  const uint32_t M = a->mesh_count;
  for (uint32_t m = 0; m < M; ++m) {
    a->cmd_out[m] = a->cmd_template[m];
    a->cmd_out[m].instanceCount = 0;
    a->cmd_out[m].firstInstance = a->mesh_base[m];
  }
  int bad = 0;
  for (uint64_t i = 0; i < a->instance_count; ++i) {
    const float* mn = a->aabb_min + 4 * i;
    const float* mx = a->aabb_max + 4 * i;
    uint32_t mesh;
    memcpy(&mesh, mn + 3, 4);
    const int vis = oracle_aabb_visible(mn, mx, a->planes);
    if (a->mask) a->mask[i] = (uint8_t)vis;
    if (!vis) continue;
    if (mesh >= M) {
      bad = 1;
      if (a->mask) a->mask[i] = 0;
      continue;
    }
    const uint32_t slot = a->mesh_base[mesh] + a->cmd_out[mesh].instanceCount++;
    a->visible[slot] = (uint32_t)i;
  }
  return bad;
}

/* Object-space AABB of a vertex list expanded to the eight corners in the
 * order of /root/reference/src/vkmodel.cpp:166-174. */
void oracle_compute_aabb_corners(const float* positions, uint64_t count, float* corners24) {
  float mn[3] = {positions[0], positions[1], positions[2]};
  float mx[3] = {positions[0], positions[1], positions[2]};
  for (uint64_t i = 0; i < count; ++i) {
    for (int c = 0; c < 3; ++c) {
      const float p = positions[3 * i + c];
      mn[c] = p < mn[c] ? p : mn[c];
      mx[c] = p > mx[c] ? p : mx[c];
    }
  }
  const float* sel[2] = {mn, mx};
  /* (mn,mn,mn),(mx,mn,mn),(mn,mx,mn),(mn,mn,mx),(mx,mx,mn),(mx,mn,mx),(mn,mx,mx),(mx,mx,mx) */
  static const int pick[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
  for (int k = 0; k < 8; ++k)
    for (int c = 0; c < 3; ++c) corners24[3 * k + c] = sel[pick[k][c]][c];
}

/* World-space AABB of an object-space AABB (8 corners) under a column-major
 * model matrix: world = model * vec4(p,1) with the row association of the
 * deferred oracle's mat*vec; min/max over the eight transformed corners. */
void oracle_world_aabb(const float* corners24, const float* model, float out_mn[3], float out_mx[3]) {
  for (int k = 0; k < 8; ++k) {
    const float x = corners24[3 * k + 0], y = corners24[3 * k + 1], z = corners24[3 * k + 2];
    for (int r = 0; r < 3; ++r) {
      const float t0 = model[0 + r] * x;
      const float t1 = model[4 + r] * y;
      const float t2 = model[8 + r] * z;
      const float w = ((t0 + t1) + t2) + model[12 + r];
      if (k == 0) {
        out_mn[r] = w;
        out_mx[r] = w;
      } else {
        out_mn[r] = w < out_mn[r] ? w : out_mn[r];
        out_mx[r] = w > out_mx[r] ? w : out_mx[r];
      }
    }
  }
}
