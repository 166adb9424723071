"""Deterministic synthetic inputs for the two passes (host side, numpy).

The reference's scene assets (SunTemple.fbx / .fscene, /root/reference/src/main.cpp:50,104) are not
available, so the G-buffer, light table and instance AABBs are generated from a counter-based
splitmix64 stream: the same (seed, shape) always yields the same bytes, on any numpy version.
Camera and projection follow the reference's defaults: position (0,3,71), forward (0,0,-1),
up (0,-1,0) (/root/reference/inc/vkcamera.hpp:62-63), FOV 70 deg, near 0.1, far 10000
(/root/reference/src/main.cpp:95); aspect is width/height as a float (the reference's integer
division yields 1.0 for its 1000x800 window).
"""
from __future__ import annotations

import importlib

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)

SEED_DEFERRED = 0x5EED0001
SEED_CULL = 0x5EED1001

DRAW_CMD_DTYPE = np.dtype([("indexCount", "<u4"), ("instanceCount", "<u4"), ("firstIndex", "<u4"),
                           ("vertexOffset", "<i4"), ("firstInstance", "<u4")])


def _mix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def u64_stream(seed: int, stream: int, n: int) -> np.ndarray:
    """n 64-bit words of stream `stream` under `seed` (counter mode: word i depends only on i)."""
    with np.errstate(over="ignore"):
        base = _mix(np.array([np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD6E8FEB86659FD93))], np.uint64))[0]
        idx = np.arange(n, dtype=np.uint64) * _GOLDEN + base
    return _mix(idx)


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 uniform in [0, 1)."""
    return (u64_stream(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


# ------------------------------------------------------------------------------------------ camera

def _hostmath():
    return importlib.import_module("vk-renderer_amd")


def reference_camera(width: int, height: int, *, pos=(0.0, 3.0, 71.0), forward=(0.0, 0.0, -1.0), up=(0.0, -1.0, 0.0),
                     fov_deg: float = 70.0, z_near: float = 0.1, z_far: float = 10000.0) -> dict:
    """View/projection matrices built with the library's host math (the caller's glm calls,
    /root/reference/src/main.cpp:95,206-209)."""
    hm = _hostmath()
    pos = np.asarray(pos, np.float32)
    center = pos + np.asarray(forward, np.float32)
    proj = hm.perspective(float(np.radians(np.float32(fov_deg))), float(np.float32(width) / np.float32(height)), z_near, z_far)
    view = hm.look_at(pos, center, np.asarray(up, np.float32))
    vp = hm.mat4_mul(proj, view)
    inv_vp = hm.mat4_inverse(vp)
    return {"pos": pos, "proj": proj, "view": view, "vp": vp, "inv_vp": inv_vp,
            "push": hm.deferred_push(pos, inv_vp, 1.0), "planes": hm.frustum_planes(vp)}


def unproject(inv_vp: np.ndarray, width: int, height: int, depth: np.ndarray) -> np.ndarray:
    """World positions (H,W,3) in float64 for the generator's own use (light box, facing normals)."""
    M = np.asarray(inv_vp, np.float64).reshape(4, 4)  # M[col][row]
    ys, xs = np.mgrid[0:height, 0:width]
    cx = (xs + 0.5) / width * 2.0 - 1.0
    cy = (ys + 0.5) / height * 2.0 - 1.0
    clip = np.stack([cx, cy, depth.astype(np.float64), np.ones_like(cx)], -1)
    world = clip @ M  # row-vector times M[col][row] == column-major M * clip
    return world[..., :3] / world[..., 3:4]


# ------------------------------------------------------------------------------------------ deferred

def make_gbuffer(width: int, height: int, seed: int, camera: dict) -> dict:
    """albedo/material (H,W,4) u8, normal (H,W,4) f16, depth (H,W) f32."""
    n = width * height
    depth = (0.90 + (0.9995 - 0.90) * uniform(seed, 1, n)).astype(np.float32).reshape(height, width)
    P = unproject(camera["inv_vp"], width, height, depth)
    V = np.asarray(camera["pos"], np.float64)[None, None, :] - P
    # unit vector uniform on the sphere
    z = 2.0 * uniform(seed, 2, n) - 1.0
    phi = 2.0 * np.pi * uniform(seed, 3, n)
    r = np.sqrt(np.maximum(1.0 - z * z, 0.0))
    nrm = np.stack([r * np.cos(phi), r * np.sin(phi), z], -1).reshape(height, width, 3)
    # the shader negates the stored normal; "faces the camera" means dot(-n, V) > 0
    facing = (uniform(seed, 4, n) < 0.9).reshape(height, width)
    d = np.sum(-nrm * V, -1)
    flip = np.where(facing, d < 0, d > 0)
    nrm = np.where(flip[..., None], -nrm, nrm)
    nrm = nrm * (0.5 + 1.5 * uniform(seed, 5, n)).reshape(height, width, 1)
    normal = np.ones((height, width, 4), np.float16)
    normal[..., :3] = nrm.astype(np.float16)

    albedo = np.full((height, width, 4), 255, np.uint8)
    albedo[..., :3] = (u64_stream(seed, 6, n * 3) >> np.uint64(56)).astype(np.uint8).reshape(height, width, 3)
    material = np.full((height, width, 4), 255, np.uint8)
    material[..., 0] = (u64_stream(seed, 7, n) >> np.uint64(56)).astype(np.uint8).reshape(height, width)
    material[..., 1] = (u64_stream(seed, 8, n) >> np.uint64(56)).astype(np.uint8).reshape(height, width)
    metal_any = (u64_stream(seed, 9, n) >> np.uint64(56)).astype(np.uint8)
    metal_bin = np.where(uniform(seed, 10, n) < 0.5, 0, 255).astype(np.uint8)
    material[..., 2] = np.where(uniform(seed, 11, n) < 0.8, metal_bin, metal_any).reshape(height, width)
    return {"albedo": albedo, "material": material, "normal": normal, "depth": depth}


def make_lights(count: int, seed: int, camera: dict, gbuf: dict, intensity: float = 50.0) -> np.ndarray:
    """(count, 12) f32 records in the reference's 48-byte layout: pos @0, color @16, rest zero.
    Positions are uniform in the world-space box spanned by the unprojected depth range."""
    h, w = gbuf["depth"].shape
    # box of the frustum slab between the generator's depth bounds (corners only: cheap at any size)
    corners = []
    M = np.asarray(camera["inv_vp"], np.float64).reshape(4, 4)
    for cz in (0.90, 0.9995):
        for cx in (-1.0, 1.0):
            for cy in (-1.0, 1.0):
                wv = np.array([cx, cy, cz, 1.0]) @ M
                corners.append(wv[:3] / wv[3])
    corners = np.array(corners)
    lo, hi = corners.min(0), corners.max(0)
    lights = np.zeros((count, 12), np.float32)
    u = uniform(seed, 20, count * 3).reshape(count, 3)
    lights[:, 0:3] = (lo + (hi - lo) * u).astype(np.float32)
    lights[:, 4:7] = (uniform(seed, 21, count * 3).reshape(count, 3) * intensity).astype(np.float32)
    return lights


def deferred_case(width: int, height: int, light_count: int, config_index: int = 0, *, seed: int | None = None) -> dict:
    seed = SEED_DEFERRED + config_index if seed is None else seed
    cam = reference_camera(width, height)
    gbuf = make_gbuffer(width, height, seed, cam)
    lights = make_lights(light_count, seed, cam, gbuf)
    return {"camera": cam, "gbuf": gbuf, "lights": lights, "push": cam["push"], "width": width, "height": height}


# ------------------------------------------------------------------------------------------ cull

def make_cull_scene(instance_count: int, mesh_count: int, seed: int, camera: dict | None = None) -> dict:
    """Contiguous per-mesh instance segments (sizes from sorted uniform cuts: some meshes get 0 or 1
    instance, some thousands), world AABBs around the camera, template commands as the reference
    builds them (/root/reference/src/vkmodel.cpp:94-98)."""
    n, m = instance_count, mesh_count
    cam_pos = np.asarray(camera["pos"] if camera else (0.0, 3.0, 71.0), np.float64)
    cuts = np.sort((uniform(seed, 1, max(m - 1, 0)) * (n + 1)).astype(np.int64))
    bounds = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    sizes = np.diff(bounds)
    mesh_base = bounds[:-1].astype(np.uint32)
    mesh_id = np.repeat(np.arange(m, dtype=np.uint32), sizes)
    centre = cam_pos[None, :] + (uniform(seed, 2, n * 3).reshape(n, 3) - 0.5) * 2000.0
    half = np.exp(np.log(0.05) + (np.log(50.0) - np.log(0.05)) * uniform(seed, 3, n * 3).reshape(n, 3))
    aabb_min = np.zeros((n, 4), np.float32)
    aabb_max = np.zeros((n, 4), np.float32)
    aabb_min[:, :3] = (centre - half).astype(np.float32)
    aabb_max[:, :3] = (centre + half).astype(np.float32)
    aabb_min[:, 3] = mesh_id.view(np.float32)
    tri = 1 + (np.arange(m, dtype=np.int64) % 997)
    cmd = np.zeros(m, DRAW_CMD_DTYPE)
    cmd["indexCount"] = (3 * tri).astype(np.uint32)
    cmd["instanceCount"] = sizes.astype(np.uint32)
    cmd["firstIndex"] = np.concatenate([[0], np.cumsum(3 * tri)[:-1]]).astype(np.uint32)
    cmd["vertexOffset"] = np.concatenate([[0], np.cumsum(tri + 2)[:-1]]).astype(np.int32)
    cmd["firstInstance"] = 0
    return {"aabb_min": aabb_min, "aabb_max": aabb_max, "mesh_base": mesh_base, "mesh_id": mesh_id,
            "cmd_template": cmd, "sizes": sizes}


def make_adversarial_cull(count: int, seed: int, camera: dict, mesh_count: int = 16) -> dict:
    """AABBs whose deciding corner (the p-vertex of one of the six planes) lies within +-2 ulp of that
    plane, so a fused multiply-add or a re-associated sum flips the verdict."""
    n = count
    planes = np.asarray(camera["planes"], np.float64).reshape(6, 4)
    # exact (binary64) inverse of the binary32 view-projection, so points land on its planes
    M = np.linalg.inv(np.asarray(camera["vp"], np.float64).reshape(4, 4).T).T
    k = (u64_stream(seed, 1, n) % np.uint64(6)).astype(np.int64)
    ndc = uniform(seed, 2, n * 3).reshape(n, 3)
    ndc[:, 0:2] = ndc[:, 0:2] * 1.8 - 0.9
    ndc[:, 2] = 0.90 + 0.0995 * ndc[:, 2]
    # pin the coordinate that plane k bounds (Vulkan clip: x,y in [-1,1], z in [0,1])
    pin_axis = np.array([0, 0, 1, 1, 2, 2])[k]
    pin_val = np.array([-1.0, 1.0, -1.0, 1.0, 0.0, 1.0])[k]
    ndc[np.arange(n), pin_axis] = pin_val
    # keep far-plane points finite and near-plane points valid
    clip = np.concatenate([ndc, np.ones((n, 1))], 1)
    world = clip @ M
    pt = (world[:, :3] / world[:, 3:4])
    finite = np.all(np.isfinite(pt), 1) & (np.abs(pt).max(1) < 1e6)
    pt[~finite] = 0.0
    pt32 = pt.astype(np.float32)
    # nudge one coordinate by -2..2 ulp
    axis = (u64_stream(seed, 3, n) % np.uint64(3)).astype(np.int64)
    steps = (u64_stream(seed, 4, n) % np.uint64(5)).astype(np.int64) - 2
    col = pt32[np.arange(n), axis].copy()
    for s in (1, 2):
        up = steps >= s
        dn = steps <= -s
        col = np.where(up, np.nextafter(col, np.float32(np.inf)), col)
        col = np.where(dn, np.nextafter(col, np.float32(-np.inf)), col)
    pt32[np.arange(n), axis] = col
    nsign = planes[k, :3] >= 0  # p-vertex takes max where the normal component is >= 0
    ext = (1e-3 + uniform(seed, 5, n * 3).reshape(n, 3)).astype(np.float32)
    aabb_min = np.zeros((n, 4), np.float32)
    aabb_max = np.zeros((n, 4), np.float32)
    aabb_max[:, :3] = np.where(nsign, pt32, pt32 + ext)
    aabb_min[:, :3] = np.where(nsign, pt32 - ext, pt32)
    sizes = np.full(mesh_count, n // mesh_count, np.int64)
    sizes[-1] += n - sizes.sum()
    mesh_base = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint32)
    mesh_id = np.repeat(np.arange(mesh_count, dtype=np.uint32), sizes)
    aabb_min[:, 3] = mesh_id.view(np.float32)
    cmd = np.zeros(mesh_count, DRAW_CMD_DTYPE)
    cmd["indexCount"] = 36
    cmd["instanceCount"] = sizes.astype(np.uint32)
    cmd["firstIndex"] = np.arange(mesh_count, dtype=np.uint32) * 36
    cmd["vertexOffset"] = np.arange(mesh_count, dtype=np.int32) * 8
    return {"aabb_min": aabb_min, "aabb_max": aabb_max, "mesh_base": mesh_base, "mesh_id": mesh_id,
            "cmd_template": cmd, "sizes": sizes}


def cull_case(instance_count: int, mesh_count: int = 1024, config_index: int = 0, *, seed: int | None = None) -> dict:
    seed = SEED_CULL + config_index if seed is None else seed
    cam = reference_camera(3840, 2160)
    scene = make_cull_scene(instance_count, mesh_count, seed, cam)
    scene["camera"] = cam
    scene["planes"] = cam["planes"]
    return scene


# ------------------------------------------------------------------------------------------ scene trees

SEED_SCENE = 0x5EED2001
MESH_VALID = 7  # has faces | has uv0 | has triangles


def make_scene_tree(node_count: int, mesh_count: int, seed: int = SEED_SCENE, *, refs_per_node: float = 1.5, invalid_every: int = 7) -> dict:
    """A random node tree standing in for an imported aiScene: node i > 0 hangs under a random earlier
    node, local transforms are rotation * scale + translation written as aiMatrix4x4 rows (a1..d4), every
    node references a few meshes; every `invalid_every`-th mesh lacks texture coordinates (skipped by
    /root/reference/src/vkmodel.cpp:91-93).  Returns nodes / meshes lists plus per-mesh vertex positions."""
    u = uniform(seed, 0, node_count * 16)
    parent = np.zeros(node_count, np.int64)
    for i in range(1, node_count):
        parent[i] = int(u[i] * i)
    children = [[] for _ in range(node_count)]
    for i in range(1, node_count):
        children[parent[i]].append(i)
    ang = uniform(seed, 1, node_count * 3) * 2 * np.pi
    scl = 0.5 + uniform(seed, 2, node_count) * 1.5
    trn = (uniform(seed, 3, node_count * 3).reshape(-1, 3) - 0.5) * 40.0
    nref = (uniform(seed, 4, node_count) * 2 * refs_per_node).astype(np.int64)
    pick = u64_stream(seed, 5, int(nref.sum()) + 1)
    nodes, k = [], 0
    for i in range(node_count):
        a, b, c = ang[3 * i:3 * i + 3]
        rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
        m = np.eye(4)
        m[:3, :3] = (rz @ ry @ rx) * scl[i]
        m[:3, 3] = trn[i]
        refs = [int(pick[k + j] % np.uint64(mesh_count)) for j in range(int(nref[i]))] if mesh_count else []
        k += int(nref[i])
        nodes.append((m.astype(np.float32).reshape(16), children[i], refs))  # row-major = a1..d4
    vcount = 3 + (u64_stream(seed, 6, mesh_count) % np.uint64(200)).astype(np.int64)
    fcount = 1 + (u64_stream(seed, 7, mesh_count) % np.uint64(300)).astype(np.int64)
    meshes = []
    for m in range(mesh_count):
        flags = MESH_VALID if (invalid_every == 0 or (m + 1) % invalid_every) else (MESH_VALID & ~2)
        meshes.append((int(vcount[m]), int(fcount[m]), m % 11, flags))
    ext = 0.05 * np.exp(uniform(seed, 8, mesh_count) * np.log(200.0))  # half-extents 0.05 .. 10
    positions = []
    for m in range(mesh_count):
        p = (uniform(seed, 100 + m, int(vcount[m]) * 3).reshape(-1, 3) - 0.5) * 2.0 * ext[m]
        positions.append(p.astype(np.float32))
    return {"nodes": nodes, "meshes": meshes, "positions": positions}


def kept_positions(tree: dict, source_mesh: np.ndarray) -> np.ndarray:
    """The packed position stream of the vertex buffer: kept meshes in table order (vkmodel.cpp:99,116)."""
    parts = [tree["positions"][int(m)] for m in source_mesh]
    return np.concatenate(parts).astype(np.float32) if parts else np.zeros((0, 3), np.float32)


def make_instance_soup(instance_count: int, mesh_count: int, seed: int = SEED_SCENE) -> dict:
    """Large flat instance set for timing the bounds kernel: contiguous per-mesh runs of random rigid
    transforms (no tree), mesh bounds log-uniform."""
    cuts = np.sort((uniform(seed, 20, mesh_count - 1) * instance_count).astype(np.int64)) if mesh_count > 1 else np.zeros(0, np.int64)
    mesh_base = np.concatenate([[0], cuts, [instance_count]]).astype(np.uint32)
    q = uniform(seed, 21, instance_count * 4).reshape(-1, 4) - 0.5
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    rot = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + z * w), 2 * (x * z - y * w),
                    2 * (x * y - z * w), 1 - 2 * (x * x + z * z), 2 * (y * z + x * w),
                    2 * (x * z + y * w), 2 * (y * z - x * w), 1 - 2 * (x * x + y * y)], axis=1)  # columns
    s = (0.5 + uniform(seed, 22, instance_count) * 1.5)[:, None]
    t = (uniform(seed, 23, instance_count * 3).reshape(-1, 3) - 0.5) * 2000.0
    mats = np.zeros((instance_count, 16), np.float32)
    mats[:, 0:3] = rot[:, 0:3] * s
    mats[:, 4:7] = rot[:, 3:6] * s
    mats[:, 8:11] = rot[:, 6:9] * s
    mats[:, 12:15] = t
    mats[:, 15] = 1.0
    ext = 0.05 * np.exp(uniform(seed, 24, mesh_count) * np.log(1000.0))
    ctr = (uniform(seed, 25, mesh_count * 3).reshape(-1, 3) - 0.5) * 4.0
    mesh_aabb = np.zeros((mesh_count, 8), np.float32)
    mesh_aabb[:, 0:3] = ctr - ext[:, None]
    mesh_aabb[:, 4:7] = ctr + ext[:, None]
    return {"mesh_base": mesh_base, "mesh_matrices": mats, "mesh_aabb": mesh_aabb}


# ------------------------------------------------------------------------------------------ analytic scenes

def make_spheres(count: int, seed: int = SEED_SCENE, *, region: float = 60.0, centre=(0.0, 3.0, 30.0)) -> np.ndarray:
    """`count` spheres scattered in front of the reference camera (48-byte vkr_sphere records)."""
    hm = _hostmath()
    s = np.zeros(count, hm.SPHERE_DTYPE)
    u = uniform(seed, 30, count * 12).reshape(-1, 12)
    s["center"] = (np.asarray(centre) + (u[:, 0:3] - 0.5) * region).astype(np.float32)
    s["radius"] = (0.5 + u[:, 3] * 4.0).astype(np.float32)
    s["albedo"] = u[:, 4:7].astype(np.float32)
    s["roughness"] = u[:, 7].astype(np.float32)
    s["metalness"] = (u[:, 8] > 0.5).astype(np.float32)
    s["occlusion"] = u[:, 9].astype(np.float32)
    return s


def fscene_text(count: int, seed: int = SEED_SCENE, *, spot_every: int = 4) -> str:
    """A .fscene-shaped JSON document: `count` lights, every `spot_every`-th one not a point light, plus
    unrelated keys of the kinds such files carry."""
    import json
    u = uniform(seed, 40, count * 12).reshape(-1, 12)
    lights = []
    for i in range(count):
        kind = "dir_light" if spot_every and (i + 1) % spot_every == 0 else "point_light"
        lights.append({"name": f"light{i}", "type": kind,
                       "intensity": [float(np.float32(u[i, 0] * 80)), float(u[i, 1] * 80), float(u[i, 2] * 80)],
                       "pos": [float((u[i, 3] - 0.5) * 100), float(u[i, 4] * 10), float((u[i, 5] - 0.5) * 100)],
                       "direction": [float(u[i, 6] - 0.5), -1.0, float(u[i, 7] - 0.5)],
                       "opening_angle": float(u[i, 8] * 360.0), "penumbra_angle": float(u[i, 9] * 30.0),
                       "extra": {"nested": [1, 2, {"deep": 'va"lue \\ end'}], "flag": True, "none": None}})
    doc = {"version": 2, "models": [{"file": "scene.fbx", "instances": [{"name": "main", "translation": [0, 0, 0]}]}],
           "lights": lights, "cameras": [{"name": "cam", "pos": [0, 3, 71]}], "user_defined": {"tone": 1.5e-3}}
    return json.dumps(doc, indent=1)
