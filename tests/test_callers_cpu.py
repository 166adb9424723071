"""CPU tests of the callers either side of the two passes: scene tables, camera / frame constants,
light-table source, attachment encodes, analytic fill.  The oracle (scene_oracle.c) is pinned by
closed-form cases; the product's host-side C++ (no GPU involved) is compared with it bit for bit."""
from __future__ import annotations

import ctypes as C
import importlib
import json
import math

import numpy as np
import pytest


@pytest.fixture(scope="module")
def scene_mod():
    return importlib.import_module("vk-renderer_amd.scene")


@pytest.fixture(scope="module")
def frame_mod():
    return importlib.import_module("vk-renderer_amd.frame")


def translation(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)  # aiMatrix4x4: a4, b4, c4 hold the translation (row-major)
    return m.reshape(16)


# ------------------------------------------------------------------------------------------ oracle KATs

def test_oracle_scene_tables_known_answer(oracle):
    """Hand-built tree (vkmodel.cpp:55-111): root(T(1,0,0)) -> A(T(0,2,0)) [mesh 0], B(T(0,0,3)) [mesh 1, mesh 0];
    A -> C(identity) [mesh 2 (no UVs: skipped), mesh 0]."""
    nodes = [(translation(1, 0, 0), [1, 2], []),
             (translation(0, 2, 0), [3], [0]),
             (translation(0, 0, 3), [], [1, 0]),
             (np.eye(4, dtype=np.float32).reshape(16), [], [2, 0])]
    meshes = [(10, 4, 5, 1), (20, 6, 7, 1), (30, 8, 9, 0)]
    t = oracle.scene_tables(nodes, meshes)
    # explicit stack: children pushed in order, so B (last) is visited before A, then A's subtree
    assert t["pairs_mesh"].tolist() == [1, 0, 0, 2, 0]
    assert t["total_instances"] == 5 and t["kept_meshes"] == 2
    # world translations: B = (1,0,3); A = (1,2,0); C = (1,2,0)
    tr = t["pairs_matrix"][:, 12:15]
    assert tr.tolist() == [[1, 0, 3], [1, 0, 3], [1, 2, 0], [1, 2, 0], [1, 2, 0]]
    # mesh 0: instances in visiting order B, A, C; mesh 1: B
    assert t["mesh_base"].tolist() == [0, 3, 4]
    assert t["mesh_matrices"][:, 12:15].tolist() == [[1, 0, 3], [1, 2, 0], [1, 2, 0], [1, 0, 3]]
    cmds = t["draw_commands"]
    assert cmds["indexCount"].tolist() == [12, 18] and cmds["instanceCount"].tolist() == [3, 1]
    assert cmds["firstIndex"].tolist() == [0, 12] and cmds["vertexOffset"].tolist() == [0, 10] and cmds["firstInstance"].tolist() == [0, 0]
    assert t["mesh_info"].tolist() == [[12, 0, 0, 5], [18, 12, 10, 7]]
    assert t["per_mesh_data"].tolist() == [[5, 0], [7, 3]]
    assert t["totals"].tolist() == [30, 30, 4] and t["source_mesh"].tolist() == [0, 1]


def test_oracle_transform_convention(oracle):
    """a1..d4 is row-major: a rotation about z by 90 degrees maps +x to +y, and child = parent * local."""
    rz = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32).reshape(16)
    nodes = [(rz, [1], []), (translation(5, 0, 0), [], [0])]
    t = oracle.scene_tables(nodes, [(3, 1, 0, 1)])
    m = t["mesh_matrices"][0].reshape(4, 4).T  # column-major -> rows x cols
    assert np.allclose(m @ np.array([0, 0, 0, 1.0]), [0, 5, 0, 1])   # translate then rotate
    assert np.allclose(m @ np.array([1, 0, 0, 1.0]), [0, 6, 0, 1])


def test_oracle_aabbs_known_answer(oracle):
    pos = np.array([[-1, -2, -3], [4, 0, 1], [0, 5, -1], [2, 2, 6]], np.float32)
    mesh = oracle.mesh_aabbs(pos, np.array([0, 4, 4], np.uint32))
    assert mesh[0].tolist() == [-1, -2, -3, 0, 4, 5, 6, 0]
    assert np.isposinf(mesh[1, 0:3]).all() and np.isneginf(mesh[1, 4:7]).all()  # no vertices
    # rotate 90 deg about z (x -> y, y -> -x), then translate by (10, 0, 0)
    m = np.array([[0, -1, 0, 10], [1, 0, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], np.float32).T.reshape(16)
    mn, mx = oracle.instance_aabbs(m[None], np.array([0, 1], np.uint32), mesh[:1])
    assert mn[0, :3].tolist() == [5, -1, -3] and mx[0, :3].tolist() == [12, 4, 6]
    assert mn[0, 3:].view(np.uint32)[0] == 0


def test_oracle_half_and_unorm_conversions(oracle):
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2**32, 400000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0, 0x80000000, 0x33000000, 0x33000001, 0x337FFFFF, 0x33800000, 0x38800000, 0x387FFFFF, 0x477FE000, 0x477FEFFF,
                        0x477FF000, 0x7F800000, 0xFF800000, 0x3F800000, 0x3F801000, 0x3F803000, 0x38000000, 0x37FFFFFF], np.uint32)
    f = np.concatenate([bits, special]).view(np.float32)
    f = f[~np.isnan(f)]
    n = (f.size // 4) * 4
    f = f[:n].reshape(1, -1, 4)
    zeros4 = np.zeros_like(f)
    out = oracle.gbuffer_encode(zeros4, zeros4, f, np.zeros(f.shape[:2], np.float32))
    with np.errstate(over="ignore"):
        want = f.astype(np.float16)
    assert np.array_equal(out["normal"].view(np.uint16), want.view(np.uint16))
    # unorm8: NaN -> 0, clamp, ties to even
    c = np.array([[[np.nan, -1.0, 2.0, 0.5 / 255], [1.5 / 255, 2.5 / 255, 254.5 / 255, 1.0], [0.0, 0.2, 0.999, 127.5 / 255], [np.inf, -np.inf, 1e-9, 0.3]]], np.float32)
    enc = oracle.gbuffer_encode(c, c, np.zeros_like(c), np.zeros(c.shape[:2], np.float32))["albedo"]
    want8 = np.rint(np.clip(np.nan_to_num(c.astype(np.float32), nan=0.0, posinf=1.0, neginf=0.0), 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(enc, want8)
    assert enc[0, 0].tolist() == [0, 0, 255, 0] and enc[0, 1, 0] == 2 and enc[0, 1, 1] == 2 and enc[0, 1, 2] == 254


def test_oracle_surface_encode_known_answers(oracle):
    """offscreen.vert:32-35 + offscreen.frag:15-16 on cases with closed-form answers, and against an independent
    binary64 numpy evaluation of the GLSL on random input."""
    def run(n, t, texel, M):
        one = lambda v: np.asarray(v, np.float32).reshape(1, 1, -1)
        nobj = np.zeros((1, 1, 4), np.float32)
        nobj[0, 0, :3] = n
        tobj = np.zeros((1, 1, 4), np.float32)
        tobj[0, 0, :3] = t
        tex = np.ones((1, 1, 4), np.float32)
        tex[0, 0, :3] = texel
        z = np.zeros((1, 1, 4), np.float32)
        return oracle.gbuffer_encode_surfaces(z, z, tex, nobj, tobj, np.zeros((1, 1), np.float32), np.asarray(M, np.float32).reshape(1, 16))["normal"][0, 0].astype(np.float32)
    I = np.eye(4, dtype=np.float32).reshape(-1)
    # flat normal map (0.5, 0.5, 1): the geometric normal, normalised; w = 1
    assert run([0, 0, 2], [3, 0, 0], [0.5, 0.5, 1.0], I).tolist() == [0, 0, 1, 1]
    # texel (1, 0.5, 0.5) -> tangent direction; (0.5, 1, 0.5) -> bitangent = cross(N, T)
    assert run([0, 0, 1], [1, 0, 0], [1.0, 0.5, 0.5], I).tolist() == [1, 0, 0, 1]
    assert run([0, 0, 1], [1, 0, 0], [0.5, 1.0, 0.5], I).tolist() == [0, 1, 0, 1]
    # a model matrix that turns x into y (column-major: column 0 = (0,1,0)), scales z by 5 and translates: the frame
    # turns with it, the scale is normalised away, the translation is ignored
    M = np.zeros((4, 4), np.float32)
    M[0] = [0, 1, 0, 0]
    M[1] = [-1, 0, 0, 0]
    M[2] = [0, 0, 5, 0]
    M[3] = [7, 8, 9, 1]
    assert run([0, 0, 1], [1, 0, 0], [1.0, 0.5, 0.5], M.reshape(-1)).tolist() == [0, 1, 0, 1]
    assert run([0, 0, 1], [1, 0, 0], [0.5, 0.5, 1.0], M.reshape(-1)).tolist() == [0, 0, 1, 1]
    # second reading of the two shaders in binary64
    rng = np.random.default_rng(11)
    h, w, K = 6, 50, 9
    nobj = rng.normal(size=(h, w, 4)).astype(np.float32)
    tobj = rng.normal(size=(h, w, 4)).astype(np.float32)
    tex = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    idx = rng.integers(0, K, (h, w)).astype(np.uint32)
    nobj[..., 3] = idx.view(np.float32)
    mats = rng.normal(size=(K, 16)).astype(np.float32)
    z = np.zeros((h, w, 4), np.float32)
    got = oracle.gbuffer_encode_surfaces(z, z, tex, nobj, tobj, np.zeros((h, w), np.float32), mats)["normal"].astype(np.float64)
    M3 = mats.astype(np.float64).reshape(K, 4, 4)[idx][..., :3, :3]          # [..., column, row]
    rot = lambda v: np.einsum("hwcr,hwc->hwr", M3, v.astype(np.float64))
    unit = lambda v: v / np.linalg.norm(v, axis=-1, keepdims=True)
    N, T = unit(rot(nobj[..., :3])), unit(rot(tobj[..., :3]))
    B = np.cross(N, T)
    m = unit(tex[..., :3].astype(np.float64) * 2 - 1)
    want = T * m[..., :1] + B * m[..., 1:2] + N * m[..., 2:3]
    assert np.abs(got[..., :3] - want).max() < 2e-3 and (got[..., 3] == 1).all()  # half precision of the attachment


def test_oracle_camera_known_answers(oracle):
    c = oracle.Camera.default()
    assert c.state().tolist() == [0, 3, 71, 0, 0, -1, 0, -1, 0]
    v = c.view()  # lookAt: the eye maps to the origin, forward to -z
    M = v.reshape(4, 4).T
    assert np.allclose(M @ np.array([0, 3, 71, 1.0]), [0, 0, 0, 1], atol=1e-5)
    assert np.allclose(M @ np.array([0, 3, 70, 1.0]), [0, 0, -1, 1], atol=1e-5)
    c.key(0, 2.5)                                  # W: +forward * 0.2 * dt
    assert np.allclose(c.state()[:3], [0, 3, 70.5])
    c.key(2, 2.5)                                  # S: back
    c.key(5, 1.0)                                  # LEFT_SHIFT: +up * 0.2
    c.key(4, 1.0)                                  # SPACE: -up
    c.key(3, 1.0)                                  # D: +cross(forward, up) = (-1, 0, 0) * 0.2
    assert np.allclose(c.state()[:3], [-0.2, 3, 71], atol=1e-6)
    before = c.state()[3:6].copy()
    c.cursor_delta(1.0 / 60.0, 0.0)                # one degree of yaw
    after = c.state()[3:6]
    assert abs(np.linalg.norm(after) - 1) < 1e-6 and abs(after[1]) < 1e-7
    assert abs(math.degrees(math.acos(float(np.dot(before, after)))) - 1.0) < 1e-3
    c.cursor_delta(0.0, 0.0)                       # zero delta: no-op
    assert np.array_equal(c.state()[3:6], after)
    for _ in range(10):                            # pitch clamps at 0.01 pi from the up vector
        c.cursor_delta(0.0, 1.0)
    up_angle = math.acos(float(np.dot(c.state()[6:9], c.state()[3:6])))
    assert 0.0099 * math.pi < up_angle < 0.99 * math.pi + 1e-3 and np.isfinite(c.state()).all()


def test_oracle_point_lights(oracle):
    rec = oracle.pack_point_lights([[1, 2, 3]], [[4, 5, 6]], [[0, -1, 0]], [90.0], [0.0])
    assert rec[0, 0:3].tolist() == [1, 2, 3] and rec[0, 4:7].tolist() == [4, 5, 6] and rec[0, 8:11].tolist() == [0, -1, 0]
    assert abs(rec[0, 3] - math.cos(math.radians(45.0))) < 1e-7 and rec[0, 7] == 1.0 and rec[0, 11] == 0.0
    text = json.dumps({"lights": [{"type": "dir_light", "direction": [0, -1, 0]},
                                  {"type": "point_light", "pos": [1, 2, 3], "intensity": [4, 5, 6], "direction": [0, -1, 0],
                                   "opening_angle": 90, "penumbra_angle": 0}]})
    assert np.array_equal(oracle.fscene_point_lights(text), rec)


def test_oracle_fill_known_answer(oracle, vkr, synth):
    """One sphere straight ahead of the reference camera: the centre pixel sees its nearest point."""
    frame = importlib.import_module("vk-renderer_amd.frame")
    cam = synth.reference_camera(65, 65)
    sph = np.zeros(1, vkr.SPHERE_DTYPE)
    sph["center"], sph["radius"], sph["albedo"] = (0.0, 3.0, 41.0), 5.0, (1.0, 0.5, 0.25)
    sph["roughness"], sph["metalness"], sph["occlusion"] = 0.4, 1.0, 0.8
    sc = frame.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], 1, ground_plane=(0.0, 0.0, 0.0, 0.0))
    g = oracle.gbuffer_fill(bytes(sc), sph, 65, 65)
    assert g["albedo"][32, 32].tolist() == [255, 128, 64, 255]
    assert g["material"][32, 32].tolist() == [204, 102, 255, 255]
    # nearest point (0, 3, 46): outward normal +z, stored negated
    assert np.allclose(g["normal"][32, 32].astype(np.float32), [0, 0, -1, 1], atol=2e-3)
    P = np.array([0, 3, 46, 1.0])
    clip = cam["vp"].reshape(4, 4).T.astype(np.float64) @ P
    assert abs(g["depth"][32, 32] - clip[2] / clip[3]) < 1e-5
    # a corner pixel misses: the clear values of main.cpp:125-129
    assert g["albedo"][0, 0].tolist() == [0, 0, 0, 255] and g["material"][0, 0].tolist() == [0, 0, 0, 255]
    assert g["normal"][0, 0].tolist() == [0, 0, 0, 1] and g["depth"][0, 0] == 1.0
    # the unprojected depth lands back on the sphere
    world = cam["inv_vp"].reshape(4, 4).T.astype(np.float64) @ np.array([0.0, 0.0, g["depth"][32, 32], 1.0])
    assert np.allclose(world[:3] / world[3], [0, 3, 46], atol=2e-2)



def test_oracle_scene_tables_agree_with_an_independent_walk(oracle, synth):
    """The node walk and table building of vkmodel.cpp:55-111 written a second time in plain Python (explicit stack,
    double-precision matrix products) on a random 60-node tree: visiting order, per-mesh instance lists, skipped
    meshes, running index / vertex totals equal the C restatement; matrices agree to binary32 rounding."""
    tree = synth.make_scene_tree(60, 11, seed=77)
    nodes, meshes = tree["nodes"], tree["meshes"]
    t = oracle.scene_tables(nodes, [(v, f, mat, int(flags == synth.MESH_VALID)) for v, f, mat, flags in meshes])
    per_mesh = [[] for _ in meshes]
    order = []
    stack = [(0, np.eye(4))]
    while stack:
        node, parent = stack.pop()                                                     # :59-60
        local = np.asarray(nodes[node][0], np.float64).reshape(4, 4)                   # rows a, b, c, d after the transpose of :65
        world = parent @ local                                                         # :66
        for m in nodes[node][2]:                                                       # :69-70
            per_mesh[m].append(world)
            order.append(m)
        for c in nodes[node][1]:                                                       # :71-72
            stack.append((c, world))
    assert t["pairs_mesh"][:len(order)].tolist() == order and t["total_instances"] == len(order)
    kept = [k for k, (v, f, mat, valid) in enumerate(meshes) if valid == synth.MESH_VALID]   # :91-93
    assert t["source_mesh"].tolist() == kept
    first_index = vertex_offset = base = 0
    for k, m in enumerate(kept):
        v, f, mat, _ = meshes[m]
        assert t["draw_commands"][k].tolist() == (3 * f, len(per_mesh[m]), first_index, vertex_offset, 0)   # :96
        assert t["mesh_info"][k].tolist() == [3 * f, first_index, vertex_offset, mat]                        # :95
        assert t["per_mesh_data"][k].tolist() == [mat, base]                                               # :108
        got = t["mesh_matrices"][base:base + len(per_mesh[m])].reshape(-1, 4, 4).transpose(0, 2, 1)       # column-major storage
        if per_mesh[m]:
            want = np.stack(per_mesh[m])
            assert np.allclose(got, want, rtol=2e-5, atol=2e-5 * np.abs(want).max())
        first_index += 3 * f
        vertex_offset += v
        base += len(per_mesh[m])
    assert t["mesh_base"].tolist()[-1] == base


def test_oracle_instance_aabbs_agree_with_numpy(oracle, synth):
    """Eight corners in the order of vkmodel.cpp:172-173 under each model matrix, elementwise binary32 with the
    association ((m0*x + m1*y) + m2*z) + m3: bit-identical minima / maxima and the mesh id in min.w."""
    soup = synth.make_instance_soup(5000, 23)
    mats = soup["mesh_matrices"].reshape(-1, 4, 4)  # column-major: mats[i, c, r]
    base = soup["mesh_base"].astype(np.int64)
    mesh_of = np.searchsorted(base[1:], np.arange(mats.shape[0]), side="right")
    lo, hi = soup["mesh_aabb"].reshape(-1, 8)[mesh_of, 0:3], soup["mesh_aabb"].reshape(-1, 8)[mesh_of, 4:7]
    corners = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    mn = np.full((mats.shape[0], 3), np.inf, np.float32)
    mx = np.full((mats.shape[0], 3), -np.inf, np.float32)
    for sx, sy, sz in corners:
        x, y, z = (hi if sx else lo)[:, 0], (hi if sy else lo)[:, 1], (hi if sz else lo)[:, 2]
        for r in range(3):
            w = ((mats[:, 0, r] * x + mats[:, 1, r] * y) + mats[:, 2, r] * z) + mats[:, 3, r]
            assert w.dtype == np.float32
            mn[:, r] = np.minimum(mn[:, r], w)
            mx[:, r] = np.maximum(mx[:, r], w)
    omn, omx = oracle.instance_aabbs(soup["mesh_matrices"], soup["mesh_base"], soup["mesh_aabb"])
    assert omn[:, :3].tobytes() == mn.tobytes() and omx[:, :3].tobytes() == mx.tobytes()
    assert np.array_equal(omn[:, 3].view(np.uint32), mesh_of.astype(np.uint32))


# ------------------------------------------------------------------------------------------ product host code vs oracle

@pytest.mark.parametrize("nodes,meshes,seed", [(1, 1, 1), (50, 12, 2), (400, 37, 3), (5, 0, 4)])
def test_scene_tables_match_oracle(vkr, synth, oracle, scene_mod, nodes, meshes, seed):
    t = synth.make_scene_tree(nodes, meshes, seed)
    tb = scene_mod.build_tables(scene_mod.SceneDescription(t["nodes"], t["meshes"]))
    o = oracle.scene_tables(t["nodes"], [(v, f, m, 1 if fl == 7 else 0) for v, f, m, fl in t["meshes"]])
    c = tb["counts"]
    assert (c["kept_meshes"], c["total_instances"]) == (o["kept_meshes"], o["total_instances"])
    assert [c["total_vertices"], c["total_indices"], c["kept_instances"]] == o["totals"].tolist()
    for k in ("mesh_base", "vertex_base", "source_mesh", "mesh_matrices"):
        assert np.array_equal(tb[k], o[k]), k
    assert tb["draw_commands"].tobytes() == o["draw_commands"].tobytes()
    assert np.array_equal(tb["mesh_info"].view(np.uint32).reshape(-1, 4), o["mesh_info"])
    assert np.array_equal(tb["per_mesh_data"].view(np.uint32).reshape(-1, 2), o["per_mesh_data"])
    # the reference's invariant: instances of a mesh are contiguous and the command's count matches
    assert np.array_equal(np.diff(tb["mesh_base"]), tb["draw_commands"]["instanceCount"])


def test_scene_description_errors(vkr, scene_mod):
    eye = np.eye(4, dtype=np.float32).reshape(16)
    with pytest.raises(vkr.VkrError, match="breaks the tree"):   # node 1 listed twice
        scene_mod.build_tables(scene_mod.SceneDescription([(eye, [1, 1], []), (eye, [], [])], [(3, 1, 0, 7)]))
    with pytest.raises(vkr.VkrError, match="breaks the tree"):   # the root as a child (a cycle)
        scene_mod.build_tables(scene_mod.SceneDescription([(eye, [1], []), (eye, [0], [])], [(3, 1, 0, 7)]))
    with pytest.raises(vkr.VkrError, match="names mesh"):
        scene_mod.build_tables(scene_mod.SceneDescription([(eye, [], [4])], [(3, 1, 0, 7)]))
    counts = vkr.SceneCounts()
    assert vkr.lib().vkr_scene_count(None, C.byref(counts)) == vkr.ERR_INVALID_ARGUMENT


def test_camera_and_frame_setup_match_oracle(vkr, oracle, frame_mod):
    cam, ocam = frame_mod.Camera(), oracle.Camera.default()
    proj = frame_mod.reference_projection(1920, 1080)
    rng = np.random.default_rng(3)
    for step in range(40):
        key, dt = int(rng.integers(0, 6)), float(np.float32(rng.uniform(0.1, 3.0)))
        dx, dy = (float(np.float32(v)) for v in rng.uniform(-0.5, 0.5, 2))
        cam.key(key, dt)
        ocam.key(key, dt)
        cam.cursor_delta(dx, dy)
        ocam.cursor_delta(dx, dy)
        assert np.array_equal(np.concatenate([cam.pos, cam.forward]), ocam.state()[:6]), step
        fc = frame_mod.frame_setup(cam, proj)
        assert np.array_equal(fc["view"], ocam.view())
        vp = oracle.mat_mul(proj, ocam.view())
        assert np.allclose(fc["view_proj"], vp, rtol=1e-6, atol=1e-6)
        assert np.array_equal(fc["planes"], oracle.frustum_planes(fc["view_proj"]))
        assert np.array_equal(fc["deferred_push"][:3], cam.pos) and fc["deferred_push"][3] == 1.0
        assert np.array_equal(fc["deferred_push"][4:], fc["inv_view_proj"])
        prod = oracle.mat_mul(fc["view_proj"], fc["inv_view_proj"]).reshape(4, 4)
        assert np.allclose(prod, np.eye(4), atol=5e-3)
    cam.key(99, 1.0)  # unknown key: ignored


def test_camera_path_step(vkr, frame_mod):
    cam = frame_mod.Camera()
    f0, p0 = cam.forward, cam.pos
    cam.step_path(0.5, 1.0)
    assert np.allclose(cam.pos - p0, 0.5 * f0, atol=1e-6)
    assert abs(math.degrees(math.acos(float(np.clip(np.dot(f0, cam.forward), -1, 1)))) - 1.0) < 1e-3


def test_fscene_parser_matches_json(vkr, synth, oracle, frame_mod):
    for count, seed in ((0, 1), (1, 2), (13, 3), (200, 4)):
        text = synth.fscene_text(count, seed)
        got = frame_mod.fscene_point_lights(text)
        want = oracle.fscene_point_lights(text)
        assert got.shape[0] == want.shape[0] == sum(1 for l in json.loads(text)["lights"] if l["type"] == "point_light")
        assert got.tobytes() == want.tobytes()
    # compact separators, exponents, integers, trailing vector elements, unicode escapes in skipped strings
    text = '{"a":"\\u00e9\\n","lights":[{"type":"point_light","pos":[1e1,-2.5E-1,3,99],"intensity":[1,2,3],"direction":[0,0,1],' \
           '"opening_angle":180,"penumbra_angle":1.5e1}],"z":[[],{}]}'
    got = frame_mod.fscene_point_lights(text)
    assert got.tobytes() == oracle.fscene_point_lights(text).tobytes() and got["pos"][0].tolist() == [10.0, -0.25, 3.0]
    assert frame_mod.fscene_point_lights("{}").shape == (0,)
    # a numeric literal far longer than any fixed buffer (all the digits of a double, then padding zeros): Python's
    # json and nlohmann::json both accept it
    long_x = "0." + "3" * 120 + "0" * 200 + "e1"
    text = '{"lights":[{"type":"point_light","pos":[' + long_x + ',2,3],"intensity":[1,2,3],"direction":[0,0,1],"opening_angle":90,"penumbra_angle":5}]}'
    got = frame_mod.fscene_point_lights(text)
    assert got.tobytes() == oracle.fscene_point_lights(text).tobytes() and got["pos"][0][0] == np.float32(float(long_x))


def test_fscene_parser_errors(vkr, frame_mod):
    n = C.c_uint32()
    for bad in ('{"lights": [', '[1,2]', '{"lights": {"a": 1}}', '{"lights": [{"type": "point_light", "pos": [1, 2]}]}', '{"lights": [}'):
        with pytest.raises(vkr.VkrError):
            frame_mod.fscene_point_lights(bad)
    with pytest.raises(vkr.VkrError, match="lacks"):
        frame_mod.fscene_point_lights('{"lights": [{"type": "point_light", "pos": [1, 2, 3]}]}')
    text = importlib.import_module("vk-renderer_amd.synth").fscene_text(8, 5).encode()
    out = np.zeros(2, vkr.LIGHT_INFO_DTYPE)
    st = vkr.lib().vkr_fscene_point_lights(text, len(text), out.ctypes.data, 2, C.byref(n))
    assert st == vkr.ERR_OUT_OF_MEMORY and n.value == 6 and b"capacity" in vkr.lib().vkr_last_error()


def test_fscene_parser_survives_hostile_documents(vkr, synth, frame_mod):
    """Untrusted input: nesting far deeper than any scene file is rejected rather than followed down the call
    stack, moderate nesting is accepted, and byte-level mutations of a valid document never crash the reader."""
    with pytest.raises(vkr.VkrError, match="nesting"):
        frame_mod.fscene_point_lights('{"a":' * 200000 + "1" + "}" * 200000)
    with pytest.raises(vkr.VkrError, match="nesting"):
        frame_mod.fscene_point_lights('{"lights": [], "x": ' + "[" * 300000 + "]" * 300000 + "}")
    assert frame_mod.fscene_point_lights('{"x": ' + "[" * 200 + "]" * 200 + ', "lights": []}').shape == (0,)
    base = synth.fscene_text(5, 3).encode()
    rng = np.random.default_rng(5)
    n = C.c_uint32()
    out = np.zeros(16, vkr.LIGHT_INFO_DTYPE)
    outcomes = set()
    for _ in range(3000):
        t = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            pos = int(rng.integers(0, len(t)))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                t[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                del t[pos:pos + int(rng.integers(1, 9))]
            elif kind == 2:
                t[pos:pos] = rng.choice([b"{", b"}", b"[", b"]", b",", b":", b'"', b"\\", b"1e999", b"\\u00", b"\x00"])
            else:
                del t[pos:]
        outcomes.add(vkr.lib().vkr_fscene_point_lights(bytes(t), len(t), out.ctypes.data, 16, C.byref(n)))
    assert outcomes <= {vkr.OK, vkr.ERR_INVALID_ARGUMENT, vkr.ERR_OUT_OF_MEMORY} and vkr.ERR_INVALID_ARGUMENT in outcomes


def test_pack_point_lights_matches_oracle(vkr, oracle, frame_mod):
    rng = np.random.default_rng(11)
    n = 1000
    pos, col, d = (rng.normal(size=(n, 3)).astype(np.float32) * 50 for _ in range(3))
    oa, pa = rng.uniform(0, 360, n).astype(np.float32), rng.uniform(0, 90, n).astype(np.float32)
    got = frame_mod.pack_point_lights(pos, col, d, oa, pa)
    assert got.tobytes() == oracle.pack_point_lights(pos, col, d, oa, pa).tobytes()
    assert got.dtype.itemsize == 48 and np.array_equal(got["pos"], pos) and np.array_equal(got["color"], col)


def test_new_struct_layouts(vkr):
    assert C.sizeof(vkr.Camera) == 36 and C.sizeof(vkr.FrameConstants) == 4 * (16 * 3 + 24 + 20)
    assert C.sizeof(vkr.GBufferScene) == 208 and vkr.GBufferScene.sphere_count.offset == 140 and vkr.GBufferScene.ground_plane.offset == 144
    assert C.sizeof(vkr.SceneCounts) == 20 and C.sizeof(vkr.SceneTables) == 56 and C.sizeof(vkr.SceneDesc) == 88
    assert vkr.SPHERE_DTYPE.itemsize == 48 and vkr.LIGHT_INFO_DTYPE.itemsize == 48
    assert C.sizeof(vkr.GBufferSurfaceArgs) == 7 * 8 + 8 + 4 * 24 and vkr.GBufferSurfaceArgs.albedo.offset == 64
    assert vkr.LIGHT_INFO_DTYPE.fields["color"][1] == 16 and vkr.LIGHT_INFO_DTYPE.fields["dir"][1] == 32


# ------------------------------------------------------------------------------------------ committed fixtures

GOLDEN = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def test_golden_scene_and_lights(vkr, synth, oracle, scene_mod, frame_mod):
    """The oracle and the product's host code both reproduce the committed fixtures byte for byte."""
    z = np.load(GOLDEN + "/scene_40nodes_9meshes.npz")
    tree = synth.make_scene_tree(40, 9, synth.SEED_SCENE)
    assert np.array_equal(np.stack([n[0] for n in tree["nodes"]]), z["node_transform"])  # the generator is stable
    o = oracle.scene_tables(tree["nodes"], [(v, f, m, 1 if fl == 7 else 0) for v, f, m, fl in tree["meshes"]])
    tb = scene_mod.build_tables(scene_mod.SceneDescription(tree["nodes"], tree["meshes"]))
    for k in ("mesh_matrices", "mesh_base", "vertex_base", "source_mesh"):
        assert np.array_equal(o[k], z[k]) and np.array_equal(tb[k], z[k]), k
    assert o["draw_commands"].tobytes() == z["draw_commands"].tobytes() == tb["draw_commands"].tobytes()
    assert np.array_equal(tb["mesh_info"].view(np.uint32).reshape(-1, 4), z["mesh_info"])
    positions = synth.kept_positions(tree, tb["source_mesh"])
    assert np.array_equal(positions, z["positions"])
    mesh = oracle.mesh_aabbs(positions, tb["vertex_base"])
    mn, mx = oracle.instance_aabbs(tb["mesh_matrices"], tb["mesh_base"], mesh)
    assert np.array_equal(mesh.view(np.uint32), z["mesh_aabb"].view(np.uint32))
    assert np.array_equal(mn.view(np.uint32), z["aabb_min"].view(np.uint32)) and np.array_equal(mx.view(np.uint32), z["aabb_max"].view(np.uint32))

    text = open(GOLDEN + "/lights_9.fscene").read()
    zl = np.load(GOLDEN + "/lights_camera.npz")
    assert oracle.fscene_point_lights(text).tobytes() == zl["light_records"].tobytes()
    assert frame_mod.fscene_point_lights(text).tobytes() == zl["light_records"].tobytes()
    cam = frame_mod.Camera()
    for want in zl["camera_path"]:
        cam.step_path(0.5, 1.0)
        assert np.array_equal(np.concatenate([cam.pos, cam.forward, cam.view()]), want)


def test_golden_gbuffer_fill(oracle):
    z = np.load(GOLDEN + "/gbuffer_fill_96x54_12spheres.npz")
    g = oracle.gbuffer_fill(z["scene"].tobytes(), z["spheres"], 96, 54)
    for k in ("albedo", "material", "depth"):
        assert np.array_equal(g[k], z[k]), k
    assert np.array_equal(g["normal"].view(np.uint16), z["normal"].view(np.uint16))
    assert 0.3 < (z["depth"] < 1.0).mean() < 1.0
