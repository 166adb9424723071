// Host-only fuzz harness for vkr_scene_count / vkr_scene_build_tables under ASan/UBSan: random trees, including
// out-of-range child / mesh indices, non-monotonic offsets, shared children and cycles.
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "vkr_c.h"
int32_t vkr_fail(int32_t code, const char* fmt, ...) { (void)fmt; return code; }
static uint64_t s = 0x1234567ull;
static uint64_t rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char** argv) {
  const long iters = atol(argv[1]);
  long ok = 0, bad = 0;
  for (long it = 0; it < iters; ++it) {
    const uint32_t n = 1 + rnd() % 40, m = rnd() % 12;
    const bool hostile = rnd() % 3 == 0;
    std::vector<float> xf(16 * n);
    for (auto& v : xf) v = (float)(rnd() % 2000) / 100.0f - 10.0f;
    // a valid tree first (node i > 0 hangs under a random earlier node), then, in hostile mode, corrupted entries
    std::vector<std::vector<uint32_t>> kids(n);
    for (uint32_t i = 1; i < n; ++i) kids[rnd() % i].push_back(i);
    std::vector<uint32_t> cb(n + 1), children, mb(n + 1), meshes;
    for (uint32_t i = 0; i < n; ++i) {
      cb[i] = (uint32_t)children.size();
      mb[i] = (uint32_t)meshes.size();
      for (uint32_t c : kids[i]) {
        if (hostile && rnd() % 16 == 0) c = rnd() % (n + 3);  // back edges (cycles), shared children, self loops, out of range
        children.push_back(c);
      }
      const uint32_t nm = m ? rnd() % 3 : 0;
      for (uint32_t k = 0; k < nm; ++k) {
        uint32_t q = rnd() % m;
        if (hostile && rnd() % 8 == 0) q = rnd() % (m + 3);
        meshes.push_back(q);
      }
    }
    cb[n] = (uint32_t)children.size();
    mb[n] = (uint32_t)meshes.size();
    if (hostile && rnd() % 6 == 0) cb[rnd() % (n + 1)] = (uint32_t)(rnd() % (children.size() + 1));   // broken offsets (within the arrays: their lengths are the caller's contract)
    if (hostile && rnd() % 6 == 0) mb[rnd() % (n + 1)] = (uint32_t)(rnd() % (meshes.size() + 1));
    std::vector<uint32_t> vc(m), fc(m), mat(m), fl(m);
    for (uint32_t i = 0; i < m; ++i) { vc[i] = 3 + rnd() % 100; fc[i] = 1 + rnd() % 100; mat[i] = rnd() % 5; fl[i] = rnd() % 8; }
    if (hostile && m && rnd() % 10 == 0) vc[rnd() % m] = 0xFFFFFFF0u;  // overflowing totals
    // exact-size heap copies so that out-of-range reads are caught
    vkr_scene_desc d{};
    d.node_count = n; d.node_transform = xf.data(); d.node_child_begin = cb.data(); d.node_children = children.data();
    d.node_mesh_begin = mb.data(); d.node_meshes = meshes.data(); d.mesh_count = m;
    d.mesh_vertex_count = vc.data(); d.mesh_face_count = fc.data(); d.mesh_material = mat.data(); d.mesh_flags = fl.data();
    vkr_scene_counts c{};
    if (vkr_scene_count(&d, &c) != VKR_OK) { ++bad; continue; }
    if (c.kept_instances > 5000000u) { ++bad; continue; }
    std::vector<vkr_mesh_info> mi(c.kept_meshes); std::vector<vkr_draw_indexed_indirect_command> dc(c.kept_meshes);
    std::vector<vkr_per_mesh_data> pm(c.kept_meshes); std::vector<float> mats((size_t)c.kept_instances * 16);
    std::vector<uint32_t> base(c.kept_meshes + 1), vb(c.kept_meshes + 1), src(c.kept_meshes);
    vkr_scene_tables t{mi.data(), dc.data(), pm.data(), mats.data(), base.data(), vb.data(), src.data()};
    if (vkr_scene_build_tables(&d, &c, &t) != VKR_OK) { ++bad; continue; }
    ++ok;
  }
  printf("ok %ld rejected %ld\n", ok, bad);
}
