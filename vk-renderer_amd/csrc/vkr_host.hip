// Host-side pieces of the callers either side of the two GPU passes: the fly camera and the per-frame
// constants (/root/reference/inc/vkcamera.hpp, /root/reference/src/main.cpp:206-209), the point-light
// table read from a .fscene file (/root/reference/src/main.cpp:48-70), and the scene tables
// vw::Scene::Scene builds (/root/reference/src/vkmodel.cpp:55-111).  Plain C++; nothing here touches the GPU.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "vkr_c.h"
#include "vkr_internal.h"
#include "vkr_math.h"

namespace {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kDegToRad = 0.01745329251994329576923690768489f;  // glm::radians' factor

inline void cross(const float a[3], const float b[3], float out[3]) {
  const float x = a[1] * b[2] - a[2] * b[1];
  const float y = a[2] * b[0] - a[0] * b[2];
  const float z = a[0] * b[1] - a[1] * b[0];
  out[0] = x;
  out[1] = y;
  out[2] = z;
}
inline float dot(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

// ---------------------------------------------------------------------------------------------------
// A JSON reader just large enough for .fscene files: values are skipped structurally, numbers are
// read with strtod (as nlohmann::json does, then narrowed to float by the caller like
// /root/reference/src/main.cpp:62-66).
struct Json {
  const char* p;
  const char* end;
  std::string err;

  void ws() {
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  }
  bool fail(const char* what) {
    if (err.empty()) err = std::string(what) + " at byte " + std::to_string(static_cast<long long>(p - start));
    return false;
  }
  const char* start;
  bool lit(char c) {
    ws();
    if (p < end && *p == c) {
      ++p;
      return true;
    }
    return false;
  }
  bool string(std::string* out) {
    ws();
    if (p >= end || *p != '"') return fail("expected string");
    ++p;
    if (out) out->clear();
    while (p < end && *p != '"') {
      if (*p == '\\') {
        if (++p >= end) return fail("unterminated escape");
        char c = *p;
        switch (c) {
          case 'n': c = '\n'; break;
          case 't': c = '\t'; break;
          case 'r': c = '\r'; break;
          case 'b': c = '\b'; break;
          case 'f': c = '\f'; break;
          case 'u':  // keep the escape's text: no key or value the reader compares contains one
            if (end - p < 5) return fail("short \\u escape");
            if (out) out->append("\\u").append(p + 1, 4);
            p += 5;
            continue;
          default: break;  // '"', '\\', '/'
        }
        if (out) out->push_back(c);
        ++p;
      } else {
        if (out) out->push_back(*p);
        ++p;
      }
    }
    if (p >= end) return fail("unterminated string");
    ++p;
    return true;
  }
  bool number(double* out) {
    ws();
    if (p >= end) return fail("expected number");
    // the whole run of number characters, however long (a double printed with all its digits plus padding zeros
    // can exceed any fixed buffer), NUL-terminated for strtod
    const char* q = p;
    while (q < end && (std::strchr("+-.eE", *q) != nullptr || (*q >= '0' && *q <= '9'))) ++q;
    if (q == p) return fail("expected number");
    const std::string literal(p, q);
    char* stop = nullptr;
    const double v = std::strtod(literal.c_str(), &stop);
    if (stop != literal.c_str() + literal.size()) return fail("malformed number");
    p = q;
    *out = v;
    return true;
  }
  // Skipping recurses once per nesting level; documents nested deeper than any scene file are rejected
  // instead of being followed down the call stack.
  static constexpr int kMaxDepth = 256;
  int depth = 0;
  struct Level {
    int& d;
    explicit Level(int& depth) : d(depth) { ++d; }
    ~Level() { --d; }
  };
  bool skip() {  // any value
    ws();
    if (p >= end) return fail("unexpected end");
    if (*p == '"') return string(nullptr);
    if (*p == '{') {
      const Level level(depth);
      if (depth > kMaxDepth) return fail("nesting deeper than 256 levels");
      ++p;
      if (lit('}')) return true;
      do {
        if (!string(nullptr)) return false;
        if (!lit(':')) return fail("expected ':'");
        if (!skip()) return false;
      } while (lit(','));
      return lit('}') || fail("expected '}'");
    }
    if (*p == '[') {
      const Level level(depth);
      if (depth > kMaxDepth) return fail("nesting deeper than 256 levels");
      ++p;
      if (lit(']')) return true;
      do {
        if (!skip()) return false;
      } while (lit(','));
      return lit(']') || fail("expected ']'");
    }
    for (const char* w : {"true", "false", "null"}) {
      const size_t n = std::strlen(w);
      if (static_cast<size_t>(end - p) >= n && std::strncmp(p, w, n) == 0) {
        p += n;
        return true;
      }
    }
    double d;
    return number(&d);
  }
  bool vec3(float out[3]) {
    if (!lit('[')) return fail("expected '['");
    for (int i = 0; i < 3; ++i) {
      double d;
      if (i && !lit(',')) return fail("expected ','");
      if (!number(&d)) return false;
      out[i] = static_cast<float>(d);
    }
    while (lit(',')) {  // longer arrays: the reference reads elements 0..2 only
      if (!skip()) return false;
    }
    return lit(']') || fail("expected ']'");
  }
};

struct ParsedLight {
  bool is_point = false;
  bool has_type = false, has_pos = false, has_col = false, has_dir = false, has_open = false, has_pen = false;
  float pos[3], col[3], dir[3], open_deg = 0, pen_deg = 0;
};

bool parse_light(Json& j, ParsedLight* L) {
  if (!j.lit('{')) return j.fail("light: expected '{'");
  if (j.lit('}')) return true;
  do {
    std::string key;
    if (!j.string(&key)) return false;
    if (!j.lit(':')) return j.fail("expected ':'");
    if (key == "type") {
      j.ws();
      if (j.p < j.end && *j.p == '"') {
        std::string v;
        if (!j.string(&v)) return false;
        L->is_point = (v == "point_light");
      } else if (!j.skip()) {
        return false;
      }
      L->has_type = true;
    } else if (key == "pos") {
      if (!j.vec3(L->pos)) return false;
      L->has_pos = true;
    } else if (key == "intensity") {
      if (!j.vec3(L->col)) return false;
      L->has_col = true;
    } else if (key == "direction") {
      if (!j.vec3(L->dir)) return false;
      L->has_dir = true;
    } else if (key == "opening_angle" || key == "penumbra_angle") {
      double d;
      j.ws();
      if (j.p < j.end && (*j.p == '-' || (*j.p >= '0' && *j.p <= '9'))) {
        if (!j.number(&d)) return false;
        (key[0] == 'o' ? L->open_deg : L->pen_deg) = static_cast<float>(d);
        (key[0] == 'o' ? L->has_open : L->has_pen) = true;
      } else if (!j.skip()) {
        return false;
      }
    } else if (!j.skip()) {
      return false;
    }
  } while (j.lit(','));
  return j.lit('}') || j.fail("light: expected '}'");
}

inline void pack_one(const float pos[3], const float col[3], const float dir[3], float open_deg, float pen_deg, vkr_light_info* o) {
  std::memcpy(o->pos, pos, 12);
  std::memcpy(o->color, col, 12);
  std::memcpy(o->dir, dir, 12);
  const float opening = open_deg * kDegToRad;  // glm::radians
  o->opening_angle = std::cos(opening / 2.0f);
  const float penumbra = pen_deg * kDegToRad;
  o->penumbra_angle = std::cos(penumbra / 2.0f);
  o->_temp = 0.0f;
}

bool mesh_kept(const vkr_scene_desc* d, uint32_t m) {
  const uint32_t need = VKR_MESH_HAS_FACES | VKR_MESH_HAS_UV0 | VKR_MESH_HAS_TRIANGLES;
  return (d->mesh_flags[m] & need) == need;
}

int32_t check_desc(const vkr_scene_desc* d) {
  if (!d) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: null");
  if (d->node_count == 0) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: no root node");
  if (!d->node_transform || !d->node_child_begin || !d->node_mesh_begin)
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: null node arrays");
  if (d->mesh_count && (!d->mesh_vertex_count || !d->mesh_face_count || !d->mesh_material || !d->mesh_flags))
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: null mesh arrays");
  if (d->node_child_begin[0] != 0 || d->node_mesh_begin[0] != 0) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: offset arrays must start at 0");
  for (uint32_t n = 0; n < d->node_count; ++n) {
    if (d->node_child_begin[n + 1] < d->node_child_begin[n] || d->node_mesh_begin[n + 1] < d->node_mesh_begin[n])
      return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: offsets of node %u decrease", n);
  }
  const uint32_t nc = d->node_child_begin[d->node_count], nm = d->node_mesh_begin[d->node_count];
  if ((nc && !d->node_children) || (nm && !d->node_meshes)) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: null index arrays");
  // every node but the root must be some node's child exactly once (a tree), or the walk would not end
  std::vector<uint8_t> seen(d->node_count, 0);
  for (uint32_t i = 0; i < nc; ++i) {
    const uint32_t c = d->node_children[i];
    if (c == 0 || c >= d->node_count || seen[c]) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: child entry %u (node %u) breaks the tree", i, c);
    seen[c] = 1;
  }
  for (uint32_t i = 0; i < nm; ++i)
    if (d->node_meshes[i] >= d->mesh_count) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene description: node mesh entry %u names mesh %u of %u", i, d->node_meshes[i], d->mesh_count);
  return VKR_OK;
}

// The walk of /root/reference/src/vkmodel.cpp:55-73: explicit stack, children pushed in order (so the
// last child is visited first), world = parent * transpose(row-major local).  visit(mesh, world) is
// called once per node-mesh reference in visiting order.
template <typename Visit>
void walk(const vkr_scene_desc* d, Visit visit) {
  struct Item {
    uint32_t node;
    float parent[16];
  };
  std::vector<Item> stack;
  Item root{0, {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  stack.push_back(root);
  while (!stack.empty()) {
    const Item it = stack.back();
    stack.pop_back();
    const float* t = d->node_transform + 16ull * it.node;
    // the constructor of :62-64 takes a1..d4 as column-major data; transpose() then makes a1..a4 row 0
    float local[16];
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) local[c * 4 + r] = t[r * 4 + c];
    Item child;
    vkrm::mul(it.parent, local, child.parent);
    for (uint32_t i = d->node_mesh_begin[it.node]; i < d->node_mesh_begin[it.node + 1]; ++i) visit(d->node_meshes[i], child.parent);
    for (uint32_t i = d->node_child_begin[it.node]; i < d->node_child_begin[it.node + 1]; ++i) {
      child.node = d->node_children[i];
      stack.push_back(child);
    }
  }
}

}  // namespace

extern "C" {

void vkr_camera_default(vkr_camera* cam) {
  if (!cam) return;
  const vkr_camera c = {{0.0f, 3.0f, 71.0f}, {0.0f, 0.0f, -1.0f}, {0.0f, -1.0f, 0.0f}};
  *cam = c;
}

void vkr_camera_view(const vkr_camera* cam, float out_view[16]) {
  if (!cam || !out_view) return;
  const float center[3] = {cam->pos[0] + cam->forward[0], cam->pos[1] + cam->forward[1], cam->pos[2] + cam->forward[2]};
  vkrm::look_at(cam->pos, center, cam->up, out_view);
}

void vkr_camera_key(vkr_camera* cam, int32_t key, float delta_time) {
  if (!cam || key < VKR_KEY_W || key > VKR_KEY_LEFT_SHIFT) return;
  float mult = 0.2f * delta_time;
  if (key == VKR_KEY_SPACE || key == VKR_KEY_A || key == VKR_KEY_S) mult *= -1.0f;
  float dir[3];
  cross(cam->forward, cam->up, dir);
  if (key == VKR_KEY_LEFT_SHIFT || key == VKR_KEY_SPACE)
    std::memcpy(dir, cam->up, sizeof(dir));
  else if (key == VKR_KEY_W || key == VKR_KEY_S)
    std::memcpy(dir, cam->forward, sizeof(dir));
  for (int i = 0; i < 3; ++i) cam->pos[i] += dir[i] * mult;
}

void vkr_camera_cursor_delta(vkr_camera* cam, float dx, float dy) {
  if (!cam) return;
  const float step = 60.0f * kDegToRad;
  const float cur_up = std::acos(dot(cam->up, cam->forward));
  float new_up = cur_up - dy * step;
  const float up_min = 0.01f * kPi, up_max = 0.99f * kPi;
  new_up = new_up < up_min ? up_min : (new_up > up_max ? up_max : new_up);
  dy = (cur_up - new_up) / step;
  const float len = std::sqrt(dx * dx + dy * dy);
  if (!(len > 0.0f)) return;  // the reference would normalise a zero axis here (NaN); a zero delta is a no-op instead
  float r[3], u[3], axis[3];
  cross(cam->forward, cam->up, r);
  cross(r, cam->forward, u);
  for (int i = 0; i < 3; ++i) axis[i] = r[i] * dx + u[i] * dy;
  cross(cam->forward, axis, axis);
  const float angle = step * len;
  // glm::rotate(mat4(1), angle, axis) applied to forward
  const float c = std::cos(angle), s = std::sin(angle);
  const float inv = 1.0f / std::sqrt(dot(axis, axis));
  const float a[3] = {axis[0] * inv, axis[1] * inv, axis[2] * inv};
  const float t[3] = {(1.0f - c) * a[0], (1.0f - c) * a[1], (1.0f - c) * a[2]};
  const float col0[3] = {c + t[0] * a[0], t[0] * a[1] + s * a[2], t[0] * a[2] - s * a[1]};
  const float col1[3] = {t[1] * a[0] - s * a[2], c + t[1] * a[1], t[1] * a[2] + s * a[0]};
  const float col2[3] = {t[2] * a[0] + s * a[1], t[2] * a[1] - s * a[0], c + t[2] * a[2]};
  const float f[3] = {cam->forward[0], cam->forward[1], cam->forward[2]};
  for (int i = 0; i < 3; ++i) cam->forward[i] = (col0[i] * f[0] + col1[i] * f[1]) + col2[i] * f[2];
}

int32_t vkr_frame_setup(const vkr_camera* cam, const float proj[16], vkr_frame_constants* out) {
  if (!cam || !proj || !out) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_frame_setup: null argument");
  vkr_camera_view(cam, out->view);
  vkrm::mul(proj, out->view, out->view_proj);
  if (!vkrm::inverse(out->view_proj, out->inv_view_proj)) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_frame_setup: proj * view is singular");
  vkrm::frustum_planes(out->view_proj, out->planes);
  std::memcpy(out->deferred_push, cam->pos, 12);
  out->deferred_push[3] = 1.0f;  // pointLightRadius as pushed at /root/reference/src/main.cpp:209
  std::memcpy(out->deferred_push + 4, out->inv_view_proj, 64);
  return VKR_OK;
}

void vkr_pack_point_lights(uint32_t count, const float* pos3, const float* intensity3, const float* direction3, const float* opening_angle_deg,
                           const float* penumbra_angle_deg, vkr_light_info* out) {
  if (!pos3 || !intensity3 || !direction3 || !opening_angle_deg || !penumbra_angle_deg || !out) return;
  for (uint32_t i = 0; i < count; ++i)
    pack_one(pos3 + 3ull * i, intensity3 + 3ull * i, direction3 + 3ull * i, opening_angle_deg[i], penumbra_angle_deg[i], out + i);
}

int32_t vkr_fscene_point_lights(const char* json_text, uint64_t json_bytes, vkr_light_info* out, uint32_t capacity, uint32_t* out_count) {
  try {
  if (!json_text || !out_count) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_fscene_point_lights: null argument");
  *out_count = 0;
  Json j{json_text, json_text + json_bytes, {}, json_text};
  uint32_t found = 0;
  bool overflow = false;
  if (!j.lit('{')) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "fscene: top level is not an object");
  bool ok = true;
  if (!j.lit('}')) {
    do {
      std::string key;
      if (!(ok = j.string(&key))) break;
      if (!(ok = j.lit(':') || j.fail("expected ':'"))) break;
      if (key != "lights") {
        if (!(ok = j.skip())) break;
        continue;
      }
      if (!(ok = j.lit('[') || j.fail("\"lights\" is not an array"))) break;
      if (j.lit(']')) continue;
      do {
        ParsedLight L;
        if (!(ok = parse_light(j, &L))) break;
        if (!L.is_point) continue;
        if (!(L.has_pos && L.has_col && L.has_dir && L.has_open && L.has_pen))
          return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "fscene: point light %u lacks pos / intensity / direction / opening_angle / penumbra_angle", found);
        if (out && found < capacity)
          pack_one(L.pos, L.col, L.dir, L.open_deg, L.pen_deg, out + found);
        else if (out)
          overflow = true;
        ++found;
      } while (j.lit(','));
      if (!ok) break;
      if (!(ok = j.lit(']') || j.fail("expected ']' after lights"))) break;
    } while (j.lit(','));
    if (ok) ok = j.lit('}') || j.fail("expected '}'");
  }
  if (!ok) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "fscene: %s", j.err.c_str());
  *out_count = found;
  if (overflow) return vkr_fail(VKR_ERR_OUT_OF_MEMORY, "fscene: %u point lights, capacity %u", found, capacity);
  return VKR_OK;
  } catch (const std::bad_alloc&) {
    return vkr_fail(VKR_ERR_OUT_OF_MEMORY, "vkr_fscene_point_lights: out of host memory");
  } catch (...) {
    return vkr_fail(VKR_ERR_DEVICE, "vkr_fscene_point_lights: unexpected exception");
  }
}

int32_t vkr_scene_count(const vkr_scene_desc* desc, vkr_scene_counts* out_counts) {
  try {
  if (!out_counts) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_scene_count: null out pointer");
  if (int32_t rc = check_desc(desc)) return rc;
  vkr_scene_counts c{};
  uint64_t vertices = 0, indices = 0, instances = 0;
  walk(desc, [&](uint32_t mesh, const float*) {
    ++c.total_instances;
    if (mesh_kept(desc, mesh)) ++instances;
  });
  for (uint32_t m = 0; m < desc->mesh_count; ++m) {
    if (!mesh_kept(desc, m)) continue;
    ++c.kept_meshes;
    vertices += desc->mesh_vertex_count[m];
    indices += 3ull * desc->mesh_face_count[m];
  }
  if (vertices > 0x7FFFFFFFull || indices > 0xFFFFFFFFull || instances > 0x7FFFFFFFull)
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "scene too large for 32-bit draw commands (%llu vertices, %llu indices, %llu instances)",
                    (unsigned long long)vertices, (unsigned long long)indices, (unsigned long long)instances);
  c.kept_instances = (uint32_t)instances;
  c.total_vertices = (uint32_t)vertices;
  c.total_indices = (uint32_t)indices;
  *out_counts = c;
  return VKR_OK;
  } catch (const std::bad_alloc&) {
    return vkr_fail(VKR_ERR_OUT_OF_MEMORY, "vkr_scene_count: out of host memory");
  } catch (...) {
    return vkr_fail(VKR_ERR_DEVICE, "vkr_scene_count: unexpected exception");
  }
}

int32_t vkr_scene_build_tables(const vkr_scene_desc* desc, const vkr_scene_counts* counts, vkr_scene_tables* out) {
  try {
  if (!counts || !out) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_scene_build_tables: null argument");
  vkr_scene_counts check;
  if (int32_t rc = vkr_scene_count(desc, &check)) return rc;
  if (std::memcmp(&check, counts, sizeof(check)) != 0) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_scene_build_tables: counts do not belong to this description");
  if (check.kept_meshes && (!out->mesh_info || !out->draw_commands || !out->per_mesh_data || !out->mesh_base || !out->vertex_base || !out->source_mesh))
    return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_scene_build_tables: null table");
  if (check.kept_instances && !out->mesh_matrices) return vkr_fail(VKR_ERR_INVALID_ARGUMENT, "vkr_scene_build_tables: null mesh_matrices");

  // :55-73  per-mesh lists of world matrices, in visiting order
  std::vector<std::vector<float>> transforms(desc->mesh_count);
  walk(desc, [&](uint32_t mesh, const float* world) { transforms[mesh].insert(transforms[mesh].end(), world, world + 16); });

  // :89-111
  uint32_t k = 0, total_vertices = 0, total_indices = 0, total_matrices = 0;
  for (uint32_t m = 0; m < desc->mesh_count; ++m) {
    if (!mesh_kept(desc, m)) continue;
    const uint32_t index_count = 3u * desc->mesh_face_count[m];
    const uint32_t instances = (uint32_t)(transforms[m].size() / 16);
    out->mesh_info[k] = {index_count, total_indices, total_vertices, desc->mesh_material[m]};
    out->draw_commands[k] = {index_count, instances, total_indices, (int32_t)total_vertices, 0u};
    out->per_mesh_data[k] = {desc->mesh_material[m], total_matrices};
    out->mesh_base[k] = total_matrices;
    out->vertex_base[k] = total_vertices;
    out->source_mesh[k] = m;
    if (instances) std::memcpy(out->mesh_matrices + 16ull * total_matrices, transforms[m].data(), transforms[m].size() * sizeof(float));
    total_vertices += desc->mesh_vertex_count[m];
    total_indices += index_count;
    total_matrices += instances;
    ++k;
  }
  if (out->mesh_base) out->mesh_base[k] = total_matrices;
  if (out->vertex_base) out->vertex_base[k] = total_vertices;
  return VKR_OK;
  } catch (const std::bad_alloc&) {
    return vkr_fail(VKR_ERR_OUT_OF_MEMORY, "vkr_scene_build_tables: out of host memory");
  } catch (...) {
    return vkr_fail(VKR_ERR_DEVICE, "vkr_scene_build_tables: unexpected exception");
  }
}

}  // extern "C"
