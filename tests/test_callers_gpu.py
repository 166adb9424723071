"""GPU parity tests of the steps either side of the two passes, through the C ABI: scene bounds
(bit-exact), attachment encodes (bit-exact), analytic G-buffer fill (bit-exact), and the frame loop
end to end (the deferred tolerance of tests/_util.py)."""
from __future__ import annotations

import ctypes as C
import importlib

import numpy as np
import pytest

from _util import assert_color_close, image_row, oracle_row

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def scene_mod():
    return importlib.import_module("vk-renderer_amd.scene")


@pytest.fixture(scope="module")
def frame_mod():
    return importlib.import_module("vk-renderer_amd.frame")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ------------------------------------------------------------------------------------------ scene bounds

@pytest.mark.parametrize("nodes,meshes,seed", [(1, 1, 1), (60, 9, 2), (3000, 300, 3)])
def test_scene_bounds_match_oracle(vkr, ctx, synth, oracle, scene_mod, nodes, meshes, seed):
    tree = synth.make_scene_tree(nodes, meshes, seed)
    tables = scene_mod.build_tables(scene_mod.SceneDescription(tree["nodes"], tree["meshes"]))
    positions = synth.kept_positions(tree, tables["source_mesh"])
    ds = scene_mod.DeviceScene(ctx, tables, positions)
    ds.compute_mesh_aabbs()
    ds.compute_instance_aabbs()
    got = ds.download()
    want_mesh = oracle.mesh_aabbs(positions, tables["vertex_base"])
    assert np.array_equal(bits(got["mesh_aabb"]), bits(want_mesh))
    want_mn, want_mx = oracle.instance_aabbs(tables["mesh_matrices"], tables["mesh_base"], want_mesh)
    assert np.array_equal(bits(got["aabb_min"]), bits(want_mn))
    assert np.array_equal(bits(got["aabb_max"]), bits(want_mx))


def test_instance_bounds_large_and_ragged(vkr, ctx, synth, oracle, scene_mod):
    """200 003 instances over 257 meshes (some empty), not a multiple of the workgroup size."""
    soup = synth.make_instance_soup(200003, 257, 9)
    soup["mesh_base"][5] = soup["mesh_base"][4]  # an empty mesh in the middle
    soup["mesh_base"] = np.sort(soup["mesh_base"])
    mats = ctx.buffer_from(soup["mesh_matrices"])
    base = ctx.buffer_from(soup["mesh_base"])
    mesh = ctx.buffer_from(soup["mesh_aabb"])
    n = soup["mesh_matrices"].shape[0]
    mn, mx = ctx.buffer(n * 16), ctx.buffer(n * 16)
    vkr._check(vkr.lib().vkr_instance_aabbs(ctx.handle, mats.ptr, base.ptr, mesh.ptr, 257, n, mn.ptr, mx.ptr))
    ctx.wait_idle()
    want_mn, want_mx = oracle.instance_aabbs(soup["mesh_matrices"], soup["mesh_base"], soup["mesh_aabb"])
    assert np.array_equal(mn.download(np.uint32), bits(want_mn).reshape(-1))
    assert np.array_equal(mx.download(np.uint32), bits(want_mx).reshape(-1))
    # the mesh id column is what the cull pass reads back as aabb_min.w
    ids = mn.download(np.uint32).reshape(-1, 4)[:, 3]
    assert np.array_equal(ids, np.searchsorted(soup["mesh_base"][1:], np.arange(n), side="right").astype(np.uint32))


def test_mesh_bounds_edge_cases(vkr, ctx, oracle):
    rng = np.random.default_rng(5)
    counts = [0, 1, 63, 64, 65, 255, 256, 257, 5000, 0, 3]
    base = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    pos = rng.normal(size=(int(base[-1]), 3)).astype(np.float32) * 100
    pos[7, 1] = np.nan           # a NaN coordinate never becomes a bound (glm::min keeps the other operand)
    pos[400, 0] = np.inf
    pos[401, 2] = -np.inf
    out = ctx.buffer(len(counts) * 32)
    pos_buf, base_buf = ctx.buffer_from(pos), ctx.buffer_from(base)
    vkr._check(vkr.lib().vkr_mesh_aabbs(ctx.handle, pos_buf.ptr, base_buf.ptr, len(counts), out.ptr))
    ctx.wait_idle()
    want = oracle.mesh_aabbs(pos, base)
    got = out.download(np.float32).reshape(-1, 8)
    # the oracle scans in order (a leading NaN would stay the bound); the position stream here has none first
    assert np.array_equal(bits(got), bits(want))
    assert np.isposinf(got[0, :3]).all() and np.isneginf(got[0, 4:7]).all()


def test_scene_to_cull_end_to_end(vkr, ctx, synth, oracle, scene_mod):
    """tables -> bounds -> vkr_prep_indirect: the command buffer equals the oracle's, built from the oracle's bounds."""
    tree = synth.make_scene_tree(2000, 150, 21)
    tables = scene_mod.build_tables(scene_mod.SceneDescription(tree["nodes"], tree["meshes"]))
    positions = synth.kept_positions(tree, tables["source_mesh"])
    ds = scene_mod.DeviceScene(ctx, tables, positions)
    ds.compute_mesh_aabbs()
    ds.compute_instance_aabbs()
    cam = synth.reference_camera(1920, 1080)
    m, n = ds.mesh_count, ds.instance_count
    cmd_template = ctx.buffer_from(tables["draw_commands"])
    cmd_out, visible = ctx.buffer(m * 20), ctx.buffer(n * 4)
    visible.fill_u32(0xFFFFFFFF)
    a = vkr.CullArgs()
    a.aabb_min, a.aabb_max, a.instance_count = ds.aabb_min.ptr, ds.aabb_max.ptr, n
    a.planes = (C.c_float * 24)(*cam["planes"].reshape(24).tolist())
    a.mesh_base, a.mesh_count = ds.mesh_base.ptr, m
    a.flags = vkr.CULL_DETERMINISTIC
    a.cmd_template, a.cmd_out, a.visible = cmd_template.ptr, cmd_out.ptr, visible.ptr
    ctx.prep_indirect(a)
    ctx.wait_idle()
    mesh = oracle.mesh_aabbs(positions, tables["vertex_base"])
    mn, mx = oracle.instance_aabbs(tables["mesh_matrices"], tables["mesh_base"], mesh)
    ref = oracle.cull(mn, mx, cam["planes"], tables["mesh_base"][:m], tables["draw_commands"])
    got = cmd_out.download(vkr.DRAW_CMD_DTYPE, m)
    assert got.tobytes() == ref["cmds"].tobytes()
    assert np.array_equal(visible.download(np.uint32, n), ref["visible"])
    assert 0 < int(got["instanceCount"].sum()) < n  # the frustum really cuts the scene
    assert np.array_equal(got["firstInstance"], tables["per_mesh_data"]["modelMatrixBaseIndex"])


def test_scene_entry_point_errors(vkr, ctx):
    lib = vkr.lib()
    buf = ctx.buffer(4096)
    assert lib.vkr_instance_aabbs(ctx.handle, buf.ptr, buf.ptr, buf.ptr, 0, 5, buf.ptr, buf.ptr) == vkr.ERR_INVALID_ARGUMENT
    assert lib.vkr_instance_aabbs(ctx.handle, buf.ptr + 4, buf.ptr, buf.ptr, 1, 5, buf.ptr, buf.ptr) == vkr.ERR_INVALID_ARGUMENT
    assert lib.vkr_instance_aabbs(ctx.handle, None, buf.ptr, buf.ptr, 1, 5, buf.ptr, buf.ptr) == vkr.ERR_INVALID_ARGUMENT
    assert lib.vkr_instance_aabbs(ctx.handle, buf.ptr, buf.ptr, buf.ptr, 1, 0, None, None) == vkr.OK
    assert lib.vkr_mesh_aabbs(ctx.handle, buf.ptr, buf.ptr, 0, None) == vkr.OK
    assert lib.vkr_mesh_aabbs(ctx.handle, buf.ptr, None, 3, buf.ptr) == vkr.ERR_INVALID_ARGUMENT


# ------------------------------------------------------------------------------------------ attachment encodes

@pytest.mark.parametrize("w,h", [(64, 8), (257, 19), (3, 5), (1000, 40)])
def test_gbuffer_encode_matches_oracle(vkr, ctx, oracle, frame_mod, w, h):
    rng = np.random.default_rng(w * 1000 + h)
    def plane():
        a = rng.uniform(-0.2, 1.2, (h, w, 4)).astype(np.float32)
        k = rng.integers(0, 256, (h, w, 4))
        tie = rng.random((h, w, 4)) < 0.3
        a[tie] = ((k[tie] + 0.5) / 255.0).astype(np.float32)   # exact and near ties of the 8-bit rounding
        a[rng.random((h, w, 4)) < 0.01] = np.nan
        return a
    albedo, material = plane(), plane()
    normal = (rng.normal(size=(h, w, 4)) * np.exp(rng.uniform(-20, 12, (h, w, 4)))).astype(np.float32)  # subnormal halves .. overflow
    normal[rng.random((h, w, 4)) < 0.01] = np.inf
    depth = rng.random((h, w)).astype(np.float32)
    gb = vkr.GBuffer(ctx, w, h)
    sources = [ctx.buffer_from(a) for a in (albedo, material, normal, depth)]
    frame_mod.gbuffer_encode(ctx, gb, *sources)
    ctx.wait_idle()
    want = oracle.gbuffer_encode(albedo, material, normal, depth)
    assert np.array_equal(gb.albedo.download(np.uint8, 4), want["albedo"])
    assert np.array_equal(gb.material.download(np.uint8, 4), want["material"])
    assert np.array_equal(gb.normal.download(np.uint16, 4), want["normal"].view(np.uint16))
    assert np.array_equal(bits(gb.depth.download(np.float32, 1)), bits(want["depth"]))


def surface_case(rng, w, h, matrix_count):
    """Random per-pixel surface samples: texels, object-space normal / tangent, a matrix index per pixel, and model
    matrices with rotation, non-uniform scale, shear and translation (the tangent frame must ignore the last)."""
    albedo = rng.uniform(-0.1, 1.1, (h, w, 4)).astype(np.float32)
    specular = rng.uniform(-0.1, 1.1, (h, w, 4)).astype(np.float32)
    normal_texel = rng.uniform(0.0, 1.0, (h, w, 4)).astype(np.float32)
    normal_texel[rng.random((h, w)) < 0.2] = np.float32([0.5, 0.5, 1.0, 1.0])  # the flat normal map
    normal_obj = rng.normal(size=(h, w, 4)).astype(np.float32)
    tangent_obj = rng.normal(size=(h, w, 4)).astype(np.float32)
    index = rng.integers(0, matrix_count, (h, w)).astype(np.uint32)
    index[rng.random((h, w)) < 0.02] = matrix_count + 5  # out of range: reads matrix 0
    normal_obj[..., 3] = index.view(np.float32)
    matrices = (rng.normal(size=(matrix_count, 16)) * np.exp(rng.uniform(-3, 3, (matrix_count, 1)))).astype(np.float32)
    depth = rng.random((h, w)).astype(np.float32)
    return albedo, specular, normal_texel, normal_obj, tangent_obj, depth, matrices


@pytest.mark.parametrize("w,h,matrix_count", [(64, 8, 3), (513, 7, 40), (3, 5, 1), (1030, 21, 500)])
def test_gbuffer_encode_surfaces_matches_oracle(vkr, ctx, oracle, frame_mod, w, h, matrix_count):
    """offscreen.vert:28-35 + offscreen.frag:13-16 per pixel: bit-exact against the C restatement, on widths either
    side of the 512-pixel workgroup span."""
    rng = np.random.default_rng(w * 77 + h)
    case = surface_case(rng, w, h, matrix_count)
    gb = vkr.GBuffer(ctx, w, h)
    bufs = [ctx.buffer_from(a) for a in case]
    frame_mod.gbuffer_encode_surfaces(ctx, gb, *bufs, matrix_count)
    ctx.wait_idle()
    want = oracle.gbuffer_encode_surfaces(*case)
    assert np.array_equal(gb.albedo.download(np.uint8, 4), want["albedo"])
    assert np.array_equal(gb.material.download(np.uint8, 4), want["material"])
    got_n, want_n = gb.normal.download(np.uint16, 4), want["normal"].view(np.uint16)
    assert np.array_equal(got_n, want_n)
    assert np.isfinite(want["normal"].astype(np.float32)).all() and (want_n[..., 3] == 0x3C00).all()  # w = 1.0
    assert np.array_equal(bits(gb.depth.download(np.float32, 1)), bits(want["depth"]))


def test_gbuffer_encode_surfaces_degenerate_and_errors(vkr, ctx, oracle, frame_mod):
    """A zero normal or tangent has no direction: NaN in both (0 / 0), at the same pixels.  Argument checks."""
    rng = np.random.default_rng(5)
    case = list(surface_case(rng, 40, 4, 2))
    case[3][1, 7, :3] = 0.0   # inNormal = 0
    case[4][2, 9, :3] = 0.0   # inTangent = 0
    case[2][3, 11, :3] = 0.5  # normal texel * 2 - 1 = 0
    gb = vkr.GBuffer(ctx, 40, 4)
    bufs = [ctx.buffer_from(a) for a in case]
    frame_mod.gbuffer_encode_surfaces(ctx, gb, *bufs, 2)
    ctx.wait_idle()
    want = oracle.gbuffer_encode_surfaces(*case)
    got = gb.normal.download(np.float16, 4)
    assert np.array_equal(np.isnan(got), np.isnan(want["normal"]))
    assert np.isnan(got[1, 7, :3]).all() and np.isnan(got[2, 9, :3]).any() and np.isnan(got[3, 11, :3]).all()
    ok = ~np.isnan(got)
    assert np.array_equal(got.view(np.uint16)[ok], want["normal"].view(np.uint16)[ok])
    with pytest.raises(vkr.VkrError, match="matrix"):
        frame_mod.gbuffer_encode_surfaces(ctx, gb, *bufs, 0)
    a = vkr.GBufferSurfaceArgs()
    with pytest.raises(vkr.VkrError):
        vkr._check(vkr.lib().vkr_gbuffer_encode_surfaces(ctx.handle, a))


# ------------------------------------------------------------------------------------------ analytic fill

def fill_case(vkr, synth, frame_mod, w, h, nspheres, seed, **scene_kw):
    cam = synth.reference_camera(w, h)
    spheres = synth.make_spheres(nspheres, seed)
    sc = frame_mod.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], nspheres, **scene_kw)
    return cam, spheres, sc


@pytest.mark.parametrize("w,h,n,kw", [(257, 131, 40, {}), (64, 64, 0, {}), (130, 70, 300, {"ground_plane": (0.0, 0.0, 0.0, 0.0)}),
                                       (96, 54, 7, {"ground_plane": (0.1, 1.0, -0.05, 2.0), "ground_tile": 1.5})])
def test_gbuffer_fill_matches_oracle(vkr, ctx, synth, oracle, frame_mod, w, h, n, kw):
    cam, spheres, sc = fill_case(vkr, synth, frame_mod, w, h, n, 30 + n, **kw)
    gb = vkr.GBuffer(ctx, w, h)
    sbuf = ctx.buffer_from(spheres) if n else ctx.buffer(64)
    frame_mod.gbuffer_fill(ctx, gb, sbuf, sc)
    ctx.wait_idle()
    want = oracle.gbuffer_fill(bytes(sc), spheres, w, h)
    assert np.array_equal(gb.albedo.download(np.uint8, 4), want["albedo"])
    assert np.array_equal(gb.material.download(np.uint8, 4), want["material"])
    assert np.array_equal(gb.normal.download(np.uint16, 4), want["normal"].view(np.uint16))
    assert np.array_equal(bits(gb.depth.download(np.float32, 1)), bits(want["depth"]))
    if n and not kw:
        hit = want["depth"] < 1.0
        assert 0.2 < hit.mean() <= 1.0  # the scene is really in view


@pytest.mark.parametrize("kind", ["tiny_far", "huge_near", "behind", "grazing", "mixed"])
def test_gbuffer_fill_with_spheres_hard_to_place(vkr, ctx, synth, oracle, frame_mod, kind):
    """The fill leaves out, per 32 x 32 tile, the spheres no ray of the tile can hit.  Spheres chosen to strain that
    decision: tiny ones far away (the binary32 discriminant of a ray gives false hits well outside such a sphere's
    true outline, and the oracle, which tests every sphere on every pixel, reports them), huge ones around and in
    front of the camera, spheres behind it, spheres whose outline runs along tile borders.  Bit-exact as ever."""
    w, h = 320, 200
    cam = synth.reference_camera(w, h)
    rng = np.random.default_rng({"tiny_far": 1, "huge_near": 2, "behind": 3, "grazing": 4, "mixed": 5}[kind])
    n = 160
    spheres = synth.make_spheres(n, 99)
    pos, fwd = np.asarray(cam["pos"], np.float64), np.array([0.0, 0.0, -1.0])
    far_points = synth.unproject(cam["inv_vp"], w, h, np.ones((h, w), np.float32))
    def put(i, centre, radius):
        spheres[i]["center"] = np.asarray(centre, np.float32)
        spheres[i]["radius"] = np.float32(radius)
    for i in range(n):
        pick = kind if kind != "mixed" else ("tiny_far", "huge_near", "behind", "grazing")[i % 4]
        side = rng.uniform(-1, 1, 2)
        if pick == "tiny_far":
            d = rng.uniform(500, 6000)
            put(i, pos + fwd * d + np.array([side[0] * d * 0.7, side[1] * d * 0.7, 0.0]), 10 ** rng.uniform(-3, 0))
        elif pick == "huge_near":
            d = rng.uniform(1, 80)
            put(i, pos + fwd * d + np.array([side[0] * d, side[1] * d, 0.0]), d * rng.uniform(0.5, 1.5))  # some contain the camera
        elif pick == "behind":
            d = rng.uniform(1, 200)
            put(i, pos - fwd * d + np.array([side[0] * d, side[1] * d, 0.0]), rng.uniform(0.1, 1.2) * d)
        else:
            # centre on the ray through a tile corner, radius a hair more or less than a pixel's width at that depth
            tx, ty = int(rng.integers(0, w // 32 + 1)) * 32, int(rng.integers(0, h // 32 + 1)) * 32
            far = far_points[min(ty, h - 1), min(tx, w - 1)]  # the pixel whose corner the tile corner is
            d = rng.uniform(0.002, 0.02)
            put(i, pos + (far - pos) * d, np.linalg.norm(far - pos) * d * rng.uniform(0.0005, 0.01))
    sc = frame_mod.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], n)
    gb = vkr.GBuffer(ctx, w, h)
    frame_mod.gbuffer_fill(ctx, gb, ctx.buffer_from(spheres), sc)
    ctx.wait_idle()
    want = oracle.gbuffer_fill(bytes(sc), spheres, w, h)
    assert np.array_equal(gb.albedo.download(np.uint8, 4), want["albedo"])
    assert np.array_equal(gb.material.download(np.uint8, 4), want["material"])
    assert np.array_equal(gb.normal.download(np.uint16, 4), want["normal"].view(np.uint16))
    assert np.array_equal(bits(gb.depth.download(np.float32, 1)), bits(want["depth"]))


def test_gbuffer_fill_bands_equal_whole(vkr, ctx, synth, frame_mod):
    w, h = 200, 99
    cam, spheres, sc = fill_case(vkr, synth, frame_mod, w, h, 25, 77)
    sbuf = ctx.buffer_from(spheres)
    whole, banded = vkr.GBuffer(ctx, w, h), vkr.GBuffer(ctx, w, h)
    frame_mod.gbuffer_fill(ctx, whole, sbuf, sc)
    for r0, r1 in ((0, 33), (33, 34), (34, 99)):
        frame_mod.gbuffer_fill(ctx, banded, sbuf, sc, row0=r0, rows=r1 - r0)
    ctx.wait_idle()
    for name, dt, ch in (("albedo", np.uint8, 4), ("material", np.uint8, 4), ("normal", np.uint16, 4), ("depth", np.uint32, 1)):
        assert np.array_equal(getattr(whole, name).download(dt, ch), getattr(banded, name).download(dt, ch)), name


def test_gbuffer_fill_errors(vkr, ctx, synth, frame_mod):
    cam, spheres, sc = fill_case(vkr, synth, frame_mod, 64, 64, 4, 1)
    gb = vkr.GBuffer(ctx, 64, 64)
    with pytest.raises(vkr.VkrError, match="null table"):
        frame_mod.gbuffer_fill(ctx, gb, None, sc)
    sc.ground_tile = 0.0
    with pytest.raises(vkr.VkrError, match="ground_tile"):
        frame_mod.gbuffer_fill(ctx, gb, ctx.buffer_from(spheres), sc)
    a = vkr.GBufferFillArgs()
    a.albedo = gb.albedo.desc()
    assert vkr.lib().vkr_gbuffer_fill(ctx.handle, C.byref(a)) == vkr.ERR_INVALID_ARGUMENT  # other planes missing


# ------------------------------------------------------------------------------------------ fill -> resolve, frame loop

def test_fill_then_resolve_matches_oracle(vkr, ctx, synth, oracle, frame_mod):
    """The deferred pass on a physically coherent G-buffer (smooth depth, unit normals, sky = clear values)."""
    w, h, L = 320, 180, 24
    cam, spheres, sc = fill_case(vkr, synth, frame_mod, w, h, 30, 5)
    gb = vkr.GBuffer(ctx, w, h)
    sbuf = ctx.buffer_from(spheres)
    frame_mod.gbuffer_fill(ctx, gb, sbuf, sc)
    ctx.wait_idle()
    host = {"albedo": gb.albedo.download(np.uint8, 4), "material": gb.material.download(np.uint8, 4),
            "normal": gb.normal.download(np.float16, 4), "depth": gb.depth.download(np.float32, 1)}
    rng = np.random.default_rng(2)
    lights = np.zeros((L, 12), np.float32)
    lights[:, 0:3] = np.array([0.0, 12.0, 30.0]) + rng.uniform(-30, 30, (L, 3)) * np.array([1, 0.3, 1])
    lights[:, 4:7] = rng.uniform(20, 400, (L, 3))
    out = ctx.image(vkr.FORMAT_R32G32B32A32_SFLOAT, w, h)
    lbuf = ctx.buffer_from(lights)
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out, lbuf, L, cam["push"]))
    ctx.wait_idle()
    got = out.download(np.float32, 4)
    ref = oracle.deferred(host, lights, cam["push"], mode="mix", want=("f64", "rgba8"))
    sky = host["depth"] >= 1.0
    assert_color_close(got[..., :3], ref["f64"][..., :3], skip_mask=ref["near_threshold"].astype(bool) | sky, what="fill->resolve")
    assert np.isnan(got[sky][:, :3]).all() == np.isnan(ref["f64"][sky][:, :3]).all()  # cleared pixels: normalize(0)
    out8 = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, w, h)
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out8, lbuf, L, cam["push"]))
    ctx.wait_idle()
    d8 = np.abs(out8.download(np.uint8, 4).astype(int) - ref["rgba8"].astype(int))
    assert d8.max() <= 1
    lit = ref["rgba8"][~sky][:, :3]
    assert lit.max() > 40 and lit.std() > 5  # the lights really light the scene


def test_full_8k_tiled_equals_whole_table_loop(vkr, ctx, synth, oracle, frame_mod):
    """BASELINE size of the tiled configuration: 7680x4320, 4096 lights, radius cutoff, on the analytic scene.
    The tile-list result (coarse bins, per-tile lists, ambient-only tiles) must be the bytes of the same kernel
    looping over the whole table, for the whole frame and for a screen band of it."""
    w, h, L = 7680, 4320, 4096
    cam, spheres, sc = fill_case(vkr, synth, frame_mod, w, h, 64, 9)
    gb = vkr.GBuffer(ctx, w, h)
    sbuf = ctx.buffer_from(spheres)
    frame_mod.gbuffer_fill(ctx, gb, sbuf, sc)
    lights = synth.deferred_case(64, 36, L, config_index=3)["lights"]
    lbuf = ctx.buffer_from(lights)
    push = np.asarray(cam["push"], np.float32).copy()
    push[3] = 25.0
    out_t, out_f = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, w, h), ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, w, h)
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out_t, lbuf, L, push, flags=vkr.DEFERRED_RADIUS_CUTOFF))
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out_f, lbuf, L, push, flags=vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS))
    ctx.wait_idle()
    tiled, flat = out_t.download(np.uint8, 4), out_f.download(np.uint8, 4)
    assert np.array_equal(tiled, flat)
    st = ctx.tile_stats()  # of the tile-list call (the whole-table call builds no lists)
    assert st["tiles"] == 240 * 135 and st["dark_tiles"] + st["listed_tiles"] + st["overflow_tiles"] == st["tiles"]
    assert st["overflow_tiles"] == 0 and 0 < st["listed_tiles"] < st["tiles"] and 0 < st["max_list"] <= st["list_cap"]
    assert 0.1 < st["mean_lights_per_pixel"] < 50
    ambient = np.rint(np.float32(0.05) * gb.albedo.download(np.uint8, 4)[..., :3].astype(np.float32)).astype(np.uint8)
    lit = (tiled[..., :3] != ambient).any(-1)
    assert 0.005 < lit.mean() < 0.9  # some of the frame is within reach of a light, most of it is not
    # rows of the frame against the oracle (binary64 lighting on the reference's binary32 positions, radius rule
    # applied per pixel): the first, the last and the three with the most lit pixels
    rows = sorted({0, h - 1} | set(np.argsort(lit.sum(1))[-3:].tolist()))
    checked_lit = 0
    for y in rows:
        planes = {"albedo": image_row(gb.albedo, y, np.uint8, 4), "material": image_row(gb.material, y, np.uint8, 4),
                  "normal": image_row(gb.normal, y, np.float16, 4), "depth": image_row(gb.depth, y, np.float32, 1)}
        ref, near = oracle_row(oracle, planes, y, h, lights, push, radius_cutoff=True)
        d = np.abs(tiled[y].astype(int) - ref.astype(int))
        d[near] = 0
        assert d.max() <= 1, f"row {y}: {int((d > 1).sum())} channels off by more than 1 LSB"
        checked_lit += int(lit[y].sum())
    assert checked_lit > 1000
    r0, r1 = 1000, 1700
    out_t.buffer.fill_u32(0)
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out_t, lbuf, L, push, flags=vkr.DEFERRED_RADIUS_CUTOFF, row0=r0, rows=r1 - r0))
    ctx.wait_idle()
    assert np.array_equal(out_t.download(np.uint8, 4)[r0:r1], flat[r0:r1])


def test_frame_loop_camera_path(vkr, ctx, synth, oracle, frame_mod):
    """Eight frames of the camera path, three in flight: every retained output equals the oracle pipeline
    (oracle fill -> oracle resolve) for the camera of the frame that wrote it, within 1 LSB."""
    w, h, L = 192, 108, 16
    spheres = synth.make_spheres(20, 11)
    rng = np.random.default_rng(4)
    lights = np.zeros((L, 12), np.float32)
    lights[:, 0:3] = np.array([0.0, 10.0, 35.0]) + rng.uniform(-25, 25, (L, 3)) * np.array([1, 0.2, 1])
    lights[:, 4:7] = rng.uniform(50, 300, (L, 3))
    scene = synth.cull_case(20000, 64)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    loop = frame_mod.FrameLoop(ctx, w, h, lights, spheres, cs, frames_in_flight=3, cull_flags=vkr.CULL_DETERMINISTIC)
    cam = frame_mod.Camera()
    assert loop.has_ready_slot()
    cams = []
    for f in range(8):
        snap = frame_mod.Camera()
        C.memmove(C.byref(snap.c), C.byref(cam.c), C.sizeof(cam.c))
        cams.append(snap)
        slot = loop.submit(cam)
        assert slot == f % 3
        cam.step_path()
    ctx.wait_idle()
    assert loop.has_ready_slot() and all(fc.signaled() for fc in loop.fences)
    proj = frame_mod.reference_projection(w, h)
    for f in (5, 6, 7):  # the frames whose outputs are still in the three slots
        fc = frame_mod.frame_setup(cams[f], proj)
        sc = frame_mod.make_gbuffer_scene(fc["view_proj"], fc["inv_view_proj"], cams[f].pos, 20)
        g = oracle.gbuffer_fill(bytes(sc), spheres, w, h)
        ref = oracle.deferred(g, lights, fc["deferred_push"], mode="mix", want=("rgba8",))
        got = loop.outputs[f % 3].download(np.uint8, 4)
        assert np.abs(got.astype(int) - ref["rgba8"].astype(int)).max() <= 1, f
    # the cull of the last frame used that frame's planes
    ref_c = oracle.cull(scene["aabb_min"], scene["aabb_max"], frame_mod.frame_setup(cams[7], proj)["planes"], scene["mesh_base"], scene["cmd_template"])
    assert cs.results()["cmds"].tobytes() == ref_c["cmds"].tobytes()
    assert not np.array_equal(loop.outputs[0].download(np.uint8, 4), loop.outputs[1].download(np.uint8, 4))  # the camera moved


def test_camera_path_frame_sharding(vkr, ctx, synth, frame_mod):
    """Frame-batch sharding: a rank that renders frames f % 2 == 1 of a 6-frame path ends with the same images and the
    same camera as the process that renders all six (its last three frames are 1, 3, 5; the full run keeps 3, 4, 5)."""
    dist = __import__("importlib").import_module("vk-renderer_amd.dist")
    w, h, L = 128, 72, 8
    spheres = synth.make_spheres(12, 3)
    lights = np.ascontiguousarray(synth.deferred_case(64, 36, L, config_index=5)["lights"], np.float32).reshape(-1, 12)

    def run(only):
        loop = frame_mod.FrameLoop(ctx, w, h, lights, spheres, None, frames_in_flight=3)
        cam = frame_mod.Camera()
        loop.run_path(cam, 6, only=only)
        return [o.download(np.uint8, 4) for o in loop.outputs], np.concatenate([np.asarray(cam.pos), np.asarray(cam.forward), cam.view()])

    full, cam_full = run(None)
    mine = dist.frame_slices(6, 2, 1)
    assert list(mine) == [1, 3, 5]
    part, cam_part = run(mine)
    assert np.array_equal(cam_full, cam_part)
    # full run: frame f sits in slot f % 3 -> slots hold 3, 4, 5; sharded run: its k-th frame sits in slot k % 3 -> 1, 3, 5
    assert np.array_equal(part[1], full[0]) and np.array_equal(part[2], full[2])
    assert not np.array_equal(part[0], full[0])


def test_camera_path_frame_at_full_size(vkr, ctx, synth, oracle, frame_mod):
    """The camera-path configuration at its real size: 2^22-instance cull, analytic 3840x2160 G-buffer fill and a
    256-light resolve per frame, three in flight.  The last frame's command buffer equals the oracle cull under that
    frame's planes byte for byte, and sampled rows of its image equal the oracle pipeline (oracle fill -> oracle
    resolve) within 1 LSB."""
    w, h, L, n, m = 3840, 2160, 256, 1 << 22, 1024
    spheres = synth.make_spheres(64)
    lights = np.ascontiguousarray(synth.deferred_case(64, 36, L, config_index=5)["lights"], np.float32).reshape(-1, 12)
    scene = synth.cull_case(n, m, config_index=2)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    loop = frame_mod.FrameLoop(ctx, w, h, lights, spheres, cs, frames_in_flight=3, cull_flags=vkr.CULL_DETERMINISTIC)
    cam = frame_mod.Camera()
    frames = 5
    for f in range(frames):
        last = frame_mod.Camera()
        C.memmove(C.byref(last.c), C.byref(cam.c), C.sizeof(cam.c))
        slot = loop.submit(cam)
        cam.step_path()
    ctx.wait_idle()
    fc = frame_mod.frame_setup(last, frame_mod.reference_projection(w, h))
    ref_c = oracle.cull(scene["aabb_min"], scene["aabb_max"], fc["planes"], scene["mesh_base"], scene["cmd_template"])
    res = cs.results()
    assert res["error"] == 0 and res["cmds"].tobytes() == ref_c["cmds"].tobytes()
    written = np.zeros(n, bool)  # slots past a mesh's count are never written (earlier frames left their own there)
    for b, c in zip(scene["mesh_base"], ref_c["cmds"]["instanceCount"]):
        written[int(b):int(b) + int(c)] = True
    assert np.array_equal(res["visible"][written], ref_c["visible"][written])
    assert 0.02 < ref_c["mask"].mean() < 0.6
    sc = frame_mod.make_gbuffer_scene(fc["view_proj"], fc["inv_view_proj"], last.pos, 64)
    got = loop.outputs[slot].download(np.uint8, 4)
    for y in (0, 700, 1080, 1500, h - 1):
        g = oracle.gbuffer_fill(bytes(sc), spheres, w, h, rows=(y, y + 1))
        planes = {k: g[k][0] for k in ("albedo", "material", "normal", "depth")}
        ref, near = oracle_row(oracle, planes, y, h, lights, fc["deferred_push"])
        d = np.abs(got[y].astype(int) - ref.astype(int))
        d[near] = 0
        assert d.max() <= 1, f"row {y}: {int((d > 1).sum())} channels off by more than 1 LSB"


# ------------------------------------------------------------------------------------------ committed fixtures

def test_golden_fixtures(vkr, ctx, scene_mod, frame_mod):
    import os
    golden = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(golden, "scene_40nodes_9meshes.npz"))
    tables = {"counts": {"kept_meshes": len(z["source_mesh"]), "kept_instances": len(z["mesh_matrices"])},
              "vertex_base": z["vertex_base"], "mesh_base": z["mesh_base"], "mesh_matrices": z["mesh_matrices"]}
    ds = scene_mod.DeviceScene(ctx, tables, z["positions"])
    ds.compute_mesh_aabbs()
    ds.compute_instance_aabbs()
    got = ds.download()
    assert np.array_equal(bits(got["mesh_aabb"]), bits(z["mesh_aabb"]))
    assert np.array_equal(bits(got["aabb_min"]), bits(z["aabb_min"])) and np.array_equal(bits(got["aabb_max"]), bits(z["aabb_max"]))

    z = np.load(os.path.join(golden, "gbuffer_fill_96x54_12spheres.npz"))
    sc = vkr.GBufferScene.from_buffer_copy(z["scene"].tobytes())
    gb = vkr.GBuffer(ctx, 96, 54)
    sbuf = ctx.buffer_from(z["spheres"])
    frame_mod.gbuffer_fill(ctx, gb, sbuf, sc)
    ctx.wait_idle()
    assert np.array_equal(gb.albedo.download(np.uint8, 4), z["albedo"]) and np.array_equal(gb.material.download(np.uint8, 4), z["material"])
    assert np.array_equal(gb.normal.download(np.uint16, 4), z["normal"].view(np.uint16))
    assert np.array_equal(bits(gb.depth.download(np.float32, 1)), bits(z["depth"]))
