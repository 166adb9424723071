"""CPU oracle for the two hot-path passes -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package; the product path never does.

The reference has no tests/golden vectors for either pass and no Vulkan driver can
run its shaders here.  The deferred restatement is pinned against the TEXT of the
reference's shader, evaluated by tests/glsl_eval.py (see deferred_oracle.c); the
cull is PARITY UNPINNED by nature - the reference's shader for it is empty (see
cull_oracle.c).  The C sources restate /root/reference/shaders/deferred.comp and
the data contract of /root/reference/src/vkmodel.cpp; this module is the ctypes
binding plus a numpy restatement of the glm host math the reference's caller
uses (/root/reference/src/main.cpp:95,206-209, inc/vkcamera.hpp:51-53).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (see Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("deferred_oracle.c", "cull_oracle.c", "scene_oracle.c", "deferred_oracle_body.inc", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class DeferredArgs(C.Structure):
    _fields_ = [
        ("albedo", C.c_void_p), ("material", C.c_void_p), ("normal", C.c_void_p), ("depth", C.c_void_p),
        ("pitch_albedo", C.c_int64), ("pitch_material", C.c_int64), ("pitch_normal", C.c_int64), ("pitch_depth", C.c_int64),
        ("width", C.c_int32), ("height", C.c_int32),
        ("lights", C.c_void_p), ("light_count", C.c_int32),
        ("push", C.c_void_p),
        ("y0", C.c_int32), ("y1", C.c_int32),
        ("flags", C.c_uint32),
        ("out_f32", C.c_void_p), ("out_f64", C.c_void_p), ("out_rgba8", C.c_void_p), ("near_threshold", C.c_void_p),
        ("view_sensitivity", C.c_void_p),
    ]


class CullArgs(C.Structure):
    _fields_ = [
        ("aabb_min", C.c_void_p), ("aabb_max", C.c_void_p), ("instance_count", C.c_uint64),
        ("planes", C.c_void_p), ("mesh_base", C.c_void_p), ("mesh_count", C.c_uint32),
        ("cmd_template", C.c_void_p), ("cmd_out", C.c_void_p), ("visible", C.c_void_p), ("mask", C.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        build()  # no-op when liboracle.so is newer than its sources
        _lib = C.CDLL(_LIB_PATH)
        for name in ("oracle_deferred_f32", "oracle_deferred_f64", "oracle_deferred_mix"):
            fn = getattr(_lib, name)
            fn.argtypes = [C.POINTER(DeferredArgs)]
            fn.restype = None
        _lib.oracle_cull.argtypes = [C.POINTER(CullArgs)]
        _lib.oracle_cull.restype = C.c_int
        _lib.oracle_compute_aabb_corners.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        _lib.oracle_compute_aabb_corners.restype = None
        _lib.oracle_world_aabb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.oracle_world_aabb.restype = None
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def deferred(gbuf: dict, lights: np.ndarray, push: np.ndarray, *, mode: str = "f32", rows=None,
             radius_cutoff: bool = False, threads: int = 1, want=("f32", "rgba8")) -> dict:
    """Evaluate the deferred resolve on the CPU.

    gbuf: dict with 'albedo' (H,W,4) u8, 'material' (H,W,4) u8, 'normal' (H,W,4) f16,
    'depth' (H,W) f32 -- arrays may be row-padded views; strides[0] is the pitch.
    lights: (L,12) f32 (48-byte records).  push: (20,) f32 (80-byte block).
    mode: 'f32' | 'f64' | 'mix'.  rows: (y0,y1) or None for the whole frame.
    Returns dict with the requested outputs (full-frame arrays; only `rows` filled)
    plus 'near_threshold'.  want may also name 'view_sensitivity': (H,W,3) f32, how far each
    channel moves when the unit view vector moves by 2^-23 across itself (the conditioning
    probe of deferred_oracle_body.inc; costs four more evaluations per pixel).
    """
    L = lib()
    H, W = gbuf["depth"].shape
    y0, y1 = rows if rows is not None else (0, H)
    lights = np.ascontiguousarray(lights, dtype=np.float32).reshape(-1, 12)
    push = np.ascontiguousarray(push, dtype=np.float32).reshape(20)
    out = {}
    out_f32 = np.zeros((H, W, 4), np.float32) if "f32" in want else None
    out_f64 = np.zeros((H, W, 4), np.float64) if "f64" in want else None
    out_u8 = np.zeros((H, W, 4), np.uint8) if "rgba8" in want else None
    near = np.zeros((H, W), np.uint8)
    sens = np.zeros((H, W, 3), np.float32) if "view_sensitivity" in want else None
    fn = getattr(L, "oracle_deferred_" + mode)

    def run(ya, yb):
        a = DeferredArgs()
        a.albedo, a.material = _ptr(gbuf["albedo"]), _ptr(gbuf["material"])
        a.normal, a.depth = _ptr(gbuf["normal"]), _ptr(gbuf["depth"])
        a.pitch_albedo, a.pitch_material = gbuf["albedo"].strides[0], gbuf["material"].strides[0]
        a.pitch_normal, a.pitch_depth = gbuf["normal"].strides[0], gbuf["depth"].strides[0]
        a.width, a.height = W, H
        a.lights, a.light_count = _ptr(lights), lights.shape[0]
        a.push = _ptr(push)
        a.y0, a.y1 = ya, yb
        a.flags = 1 if radius_cutoff else 0
        a.out_f32, a.out_f64, a.out_rgba8 = _ptr(out_f32), _ptr(out_f64), _ptr(out_u8)
        a.near_threshold = _ptr(near)
        a.view_sensitivity = _ptr(sens)
        fn(C.byref(a))

    if threads <= 1 or (y1 - y0) < 2 * threads:
        run(y0, y1)
    else:
        cuts = np.linspace(y0, y1, threads * 4 + 1).astype(int)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(lambda ab: run(int(ab[0]), int(ab[1])), zip(cuts[:-1], cuts[1:])))
    if out_f32 is not None:
        out["f32"] = out_f32
    if out_f64 is not None:
        out["f64"] = out_f64
    if out_u8 is not None:
        out["rgba8"] = out_u8
    out["near_threshold"] = near
    if sens is not None:
        out["view_sensitivity"] = sens
    return out


DRAW_CMD_DTYPE = np.dtype([("indexCount", "<u4"), ("instanceCount", "<u4"), ("firstIndex", "<u4"),
                           ("vertexOffset", "<i4"), ("firstInstance", "<u4")])
assert DRAW_CMD_DTYPE.itemsize == 20


def cull(aabb_min: np.ndarray, aabb_max: np.ndarray, planes: np.ndarray, mesh_base: np.ndarray,
         cmd_template: np.ndarray, *, want_mask: bool = True) -> dict:
    """Sequential cull + compaction.  Returns cmds (M,) DRAW_CMD_DTYPE, visible (N,) u32
    (slots past each mesh's count hold 0xFFFFFFFF), mask (N,) u8, bad (mesh id out of range)."""
    L = lib()
    aabb_min = np.ascontiguousarray(aabb_min, np.float32).reshape(-1, 4)
    aabb_max = np.ascontiguousarray(aabb_max, np.float32).reshape(-1, 4)
    n = aabb_min.shape[0]
    planes = np.ascontiguousarray(planes, np.float32).reshape(6, 4)
    mesh_base = np.ascontiguousarray(mesh_base, np.uint32)
    m = mesh_base.shape[0]
    cmd_template = np.ascontiguousarray(cmd_template, DRAW_CMD_DTYPE)
    cmds = np.zeros(m, DRAW_CMD_DTYPE)
    visible = np.full(n, 0xFFFFFFFF, np.uint32)
    mask = np.zeros(n, np.uint8) if want_mask else None
    a = CullArgs(_ptr(aabb_min), _ptr(aabb_max), n, _ptr(planes), _ptr(mesh_base), m,
                 _ptr(cmd_template), _ptr(cmds), _ptr(visible), _ptr(mask))
    bad = L.oracle_cull(C.byref(a))
    return {"cmds": cmds, "visible": visible, "mask": mask, "bad": bool(bad)}


def compute_aabb_corners(positions: np.ndarray) -> np.ndarray:
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    out = np.zeros((8, 3), np.float32)
    lib().oracle_compute_aabb_corners(_ptr(positions), positions.shape[0], _ptr(out))
    return out


def world_aabb(corners: np.ndarray, model: np.ndarray):
    corners = np.ascontiguousarray(corners, np.float32).reshape(8, 3)
    model = np.ascontiguousarray(model, np.float32).reshape(16)
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    lib().oracle_world_aabb(_ptr(corners), _ptr(model), _ptr(mn), _ptr(mx))
    return mn, mx


# ---------------------------------------------------------------------------------------------
# Host math the reference's caller takes from glm (third-party, un-vendored: the glm
# submodule directory is empty and no version is pinned).  These restate the published
# formulas of glm::perspective (RH, depth -1..1), glm::lookAt (RH) and a cofactor inverse,
# all column-major (m[col*4+row]) in binary32.  They produce *inputs* to the hot path.
# ---------------------------------------------------------------------------------------------

def perspective(fovy_rad: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    f32 = np.float32
    t = f32(np.tan(f32(fovy_rad) / f32(2)))
    m = np.zeros((4, 4), np.float32)  # m[col][row]
    m[0][0] = f32(1) / (f32(aspect) * t)
    m[1][1] = f32(1) / t
    m[2][2] = -(f32(z_far) + f32(z_near)) / (f32(z_far) - f32(z_near))
    m[2][3] = -f32(1)
    m[3][2] = -(f32(2) * f32(z_far) * f32(z_near)) / (f32(z_far) - f32(z_near))
    return m.reshape(16)


def look_at(eye, center, up) -> np.ndarray:
    eye, center, up = (np.asarray(v, np.float32) for v in (eye, center, up))

    def norm(v):
        return v / np.sqrt(np.dot(v, v), dtype=np.float32)

    f = norm(center - eye)
    s = norm(np.cross(f, up).astype(np.float32))
    u = np.cross(s, f).astype(np.float32)
    m = np.zeros((4, 4), np.float32)
    m[0][0], m[1][0], m[2][0] = s
    m[0][1], m[1][1], m[2][1] = u
    m[0][2], m[1][2], m[2][2] = -f
    m[3][0] = -np.dot(s, eye)
    m[3][1] = -np.dot(u, eye)
    m[3][2] = np.dot(f, eye)
    m[3][3] = 1
    return m.reshape(16)


def mat_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """Column-major a*b in binary32."""
    A = np.asarray(a, np.float32).reshape(4, 4).T  # rows x cols
    B = np.asarray(b, np.float32).reshape(4, 4).T
    return (A @ B).astype(np.float32).T.reshape(16)


def inverse(m: np.ndarray) -> np.ndarray:
    """Reference inverse in binary64, rounded to binary32 (for tolerance checks of the
    product's fp32 cofactor inverse)."""
    M = np.asarray(m, np.float64).reshape(4, 4).T
    return np.linalg.inv(M).T.astype(np.float32).reshape(16)


def frustum_planes(vp: np.ndarray) -> np.ndarray:
    """Six un-normalised planes of a column-major view-projection for Vulkan clip rules
    (-w<=x,y<=w, 0<=z<=w): row3+row0, row3-row0, row3+row1, row3-row1, row2, row3-row2."""
    M = np.asarray(vp, np.float32).reshape(4, 4)  # M[col][row]
    row = [M[:, r].copy() for r in range(4)]
    return np.stack([row[3] + row[0], row[3] - row[0], row[3] + row[1], row[3] - row[1],
                     row[2], row[3] - row[2]]).astype(np.float32)


# ---------------------------------------------------------------------------------------------
# Callers either side of the passes (scene_oracle.c): scene tables, bounds, camera, light
# records, attachment encodes, analytic G-buffer fill.
# ---------------------------------------------------------------------------------------------

class _Scene(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("node_transform", C.c_void_p), ("node_child_begin", C.c_void_p),
                ("node_children", C.c_void_p), ("node_mesh_begin", C.c_void_p), ("node_meshes", C.c_void_p),
                ("mesh_count", C.c_uint32), ("mesh_vertex_count", C.c_void_p), ("mesh_face_count", C.c_void_p),
                ("mesh_material", C.c_void_p), ("mesh_valid", C.c_void_p)]


def scene_tables(nodes, meshes) -> dict:
    """nodes: [(a1..d4 row-major 4x4, [children], [meshes])], node 0 = root; meshes: [(vertices, faces,
    material, valid)].  Returns what vw::Scene::Scene builds (vkmodel.cpp:55-111)."""
    L = lib()
    nt = np.ascontiguousarray([np.asarray(n[0], np.float32).reshape(16) for n in nodes], np.float32)
    cb, ch, mb, nm = [0], [], [0], []
    for _, c_, m_ in nodes:
        ch += list(c_)
        cb.append(len(ch))
        nm += list(m_)
        mb.append(len(nm))
    cb, ch, mb, nm = (np.asarray(a, np.uint32) for a in (cb, ch, mb, nm))
    ms = np.asarray(meshes, np.uint32).reshape(-1, 4)
    cols = [np.ascontiguousarray(ms[:, k]) for k in range(4)]
    s = _Scene(len(nodes), _ptr(nt), _ptr(cb), _ptr(ch), _ptr(mb), _ptr(nm), ms.shape[0], *(_ptr(c_) for c_ in cols))
    pairs_mesh = np.zeros(max(len(nm), 1), np.uint32)
    pairs_mat = np.zeros((max(len(nm), 1), 16), np.float32)
    L.oracle_scene_flatten.restype = C.c_uint32
    L.oracle_scene_flatten.argtypes = [C.POINTER(_Scene), C.c_void_p, C.c_void_p]
    n = L.oracle_scene_flatten(C.byref(s), _ptr(pairs_mesh), _ptr(pairs_mat))
    M = ms.shape[0]
    mesh_info = np.zeros((M, 4), np.uint32)
    draw = np.zeros((M, 5), np.uint32)
    per_mesh = np.zeros((M, 2), np.uint32)
    mats = np.zeros((max(n, 1), 16), np.float32)
    mesh_base = np.zeros(M + 1, np.uint32)
    vertex_base = np.zeros(M + 1, np.uint32)
    source = np.zeros(M, np.uint32)
    totals = np.zeros(3, np.uint32)
    L.oracle_scene_tables.restype = C.c_uint32
    L.oracle_scene_tables.argtypes = [C.POINTER(_Scene)] + [C.c_void_p] * 2 + [C.c_uint32] + [C.c_void_p] * 8
    k = L.oracle_scene_tables(C.byref(s), _ptr(pairs_mesh), _ptr(pairs_mat), n, _ptr(mesh_info), _ptr(draw), _ptr(per_mesh), _ptr(mats),
                              _ptr(mesh_base), _ptr(vertex_base), _ptr(source), _ptr(totals))
    return {"pairs_mesh": pairs_mesh[:n], "pairs_matrix": pairs_mat[:n], "total_instances": int(n), "kept_meshes": int(k),
            "mesh_info": mesh_info[:k], "draw_commands": draw[:k].copy().view(DRAW_CMD_DTYPE).reshape(-1), "per_mesh_data": per_mesh[:k],
            "mesh_matrices": mats[:int(totals[2])], "mesh_base": mesh_base[:k + 1], "vertex_base": vertex_base[:k + 1],
            "source_mesh": source[:k], "totals": totals}


def mesh_aabbs(positions: np.ndarray, vertex_base: np.ndarray) -> np.ndarray:
    positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
    vertex_base = np.ascontiguousarray(vertex_base, np.uint32)
    m = vertex_base.shape[0] - 1
    out = np.zeros((m, 8), np.float32)
    fn = lib().oracle_mesh_aabbs
    fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    fn(_ptr(positions), _ptr(vertex_base), m, _ptr(out))
    return out


def instance_aabbs(matrices: np.ndarray, mesh_base: np.ndarray, mesh_aabb: np.ndarray):
    matrices = np.ascontiguousarray(matrices, np.float32).reshape(-1, 16)
    mesh_base = np.ascontiguousarray(mesh_base, np.uint32)
    mesh_aabb = np.ascontiguousarray(mesh_aabb, np.float32).reshape(-1, 8)
    n = matrices.shape[0]
    mn, mx = np.zeros((n, 4), np.float32), np.zeros((n, 4), np.float32)
    fn = lib().oracle_instance_aabbs
    fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    fn(_ptr(matrices), _ptr(mesh_base), _ptr(mesh_aabb), mesh_base.shape[0] - 1, _ptr(mn), _ptr(mx))
    return mn, mx


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3)]

    @staticmethod
    def default() -> "Camera":
        c = Camera()
        lib().oracle_camera_default(C.byref(c))
        return c

    def view(self) -> np.ndarray:
        m = np.zeros(16, np.float32)
        lib().oracle_camera_view(C.byref(self), _ptr(m))
        return m

    def key(self, key: int, dt: float):
        fn = lib().oracle_camera_key
        fn.argtypes = [C.c_void_p, C.c_int, C.c_float]
        fn(C.byref(self), key, dt)

    def cursor_delta(self, dx: float, dy: float):
        fn = lib().oracle_camera_cursor_delta
        fn.argtypes = [C.c_void_p, C.c_float, C.c_float]
        fn(C.byref(self), dx, dy)

    def state(self) -> np.ndarray:
        return np.array(list(self.pos) + list(self.forward) + list(self.up), np.float32)


def pack_point_lights(pos, intensity, direction, opening_deg, penumbra_deg) -> np.ndarray:
    pos, intensity, direction = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (pos, intensity, direction))
    oa, pa = (np.ascontiguousarray(a, np.float32).reshape(-1) for a in (opening_deg, penumbra_deg))
    out = np.zeros((pos.shape[0], 12), np.float32)
    fn = lib().oracle_pack_point_lights
    fn.restype, fn.argtypes = None, [C.c_uint32] + [C.c_void_p] * 6
    fn(pos.shape[0], _ptr(pos), _ptr(intensity), _ptr(direction), _ptr(oa), _ptr(pa), _ptr(out))
    return out


def fscene_point_lights(text: str) -> np.ndarray:
    """main.cpp:48-70 with Python's json in the place of nlohmann::json: doubles narrowed to float."""
    import json
    scene = json.loads(text)
    pts = [l for l in scene.get("lights", []) if l.get("type") == "point_light"]
    if not pts:
        return np.zeros((0, 12), np.float32)
    f32 = lambda key, n: np.array([[np.float32(l[key][i]) for i in range(n)] for l in pts], np.float32)
    oa = np.array([np.float32(l["opening_angle"]) for l in pts], np.float32)
    pa = np.array([np.float32(l["penumbra_angle"]) for l in pts], np.float32)
    return pack_point_lights(f32("pos", 3), f32("intensity", 3), f32("direction", 3), oa, pa)


def gbuffer_encode(albedo, material, normal, depth) -> dict:
    """float planes (H,W,4)/(H,W) -> the four attachment formats."""
    albedo, material, normal = (np.ascontiguousarray(a, np.float32) for a in (albedo, material, normal))
    depth = np.ascontiguousarray(depth, np.float32)
    H, W = depth.shape
    out = {"albedo": np.zeros((H, W, 4), np.uint8), "material": np.zeros((H, W, 4), np.uint8),
           "normal": np.zeros((H, W, 4), np.uint16), "depth": np.zeros((H, W), np.float32)}
    fn = lib().oracle_gbuffer_encode
    fn.restype, fn.argtypes = None, [C.c_void_p] * 4 + [C.c_uint64] + [C.c_void_p] * 4
    fn(_ptr(albedo), _ptr(material), _ptr(normal), _ptr(depth), H * W, _ptr(out["albedo"]), _ptr(out["material"]), _ptr(out["normal"]), _ptr(out["depth"]))
    out["normal"] = out["normal"].view(np.float16)
    return out


def gbuffer_encode_surfaces(albedo_texel, specular_texel, normal_texel, normal_obj, tangent_obj, depth, matrices) -> dict:
    """Per-pixel offscreen.vert:28-35 + offscreen.frag:13-16, then the attachment conversions.  Planes (H,W,4) /
    (H,W); normal_obj[..., 3] carries the matrix index as uint32 bits; matrices (K,16) column-major."""
    planes = [np.ascontiguousarray(a, np.float32) for a in (albedo_texel, specular_texel, normal_texel, normal_obj, tangent_obj)]
    depth = np.ascontiguousarray(depth, np.float32)
    matrices = np.ascontiguousarray(matrices, np.float32).reshape(-1, 16)
    H, W = depth.shape
    out = {"albedo": np.zeros((H, W, 4), np.uint8), "material": np.zeros((H, W, 4), np.uint8),
           "normal": np.zeros((H, W, 4), np.uint16), "depth": np.zeros((H, W), np.float32)}
    fn = lib().oracle_gbuffer_encode_surfaces
    fn.restype, fn.argtypes = None, [C.c_void_p] * 7 + [C.c_uint32, C.c_uint64] + [C.c_void_p] * 4
    fn(*[_ptr(a) for a in planes], _ptr(depth), _ptr(matrices), matrices.shape[0], H * W,
       _ptr(out["albedo"]), _ptr(out["material"]), _ptr(out["normal"]), _ptr(out["depth"]))
    out["normal"] = out["normal"].view(np.float16)
    return out


class GBufferScene(C.Structure):
    _fields_ = [("view_proj", C.c_float * 16), ("inv_view_proj", C.c_float * 16), ("camera_pos", C.c_float * 3),
                ("sphere_count", C.c_uint32), ("ground_plane", C.c_float * 4), ("ground_albedo_a", C.c_float * 3),
                ("ground_tile", C.c_float), ("ground_albedo_b", C.c_float * 3), ("ground_roughness", C.c_float),
                ("ground_metalness", C.c_float), ("pad", C.c_float * 3)]


def gbuffer_fill(scene_bytes: bytes, spheres: np.ndarray, width: int, height: int, rows=None) -> dict:
    """scene_bytes: the 208-byte scene block (same layout as the product's vkr_gbuffer_scene)."""
    sc = GBufferScene.from_buffer_copy(scene_bytes)
    spheres = np.ascontiguousarray(spheres).view(np.uint8).reshape(-1)
    y0, y1 = rows if rows is not None else (0, height)
    h = y1 - y0
    out = {"albedo": np.zeros((h, width, 4), np.uint8), "material": np.zeros((h, width, 4), np.uint8),
           "normal": np.zeros((h, width, 4), np.uint16), "depth": np.zeros((h, width), np.float32)}
    fn = lib().oracle_gbuffer_fill
    fn.restype, fn.argtypes = None, [C.c_void_p, C.c_void_p] + [C.c_uint32] * 4 + [C.c_void_p] * 4
    fn(C.byref(sc), _ptr(spheres) if spheres.size else None, width, h, y0, height, _ptr(out["albedo"]), _ptr(out["material"]),
       _ptr(out["normal"]), _ptr(out["depth"]))
    out["normal"] = out["normal"].view(np.float16)
    return out
