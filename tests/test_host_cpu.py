"""CPU-only checks: the C-ABI library loads and exports what include/vkr_c.h declares, host math,
synthetic-input determinism, sharding helpers (including a 2-process gloo run), and the
no-fused-multiply-add guarantee of the cull object code.  No GPU compute is called here."""
import ctypes as C
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(vkr):
    header = open(os.path.join(ROOT, "include", "vkr_c.h")).read()
    declared = set(re.findall(r"VKR_API\s+[\w\s\*]+?\b(vkr_\w+)\s*\(", header))
    assert len(declared) >= 35
    lib = vkr.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in vkr_c.h but not exported"
    assert declared == set(vkr.EXPORTS), (declared ^ set(vkr.EXPORTS))
    nm = subprocess.run(["nm", "-D", "--defined-only", vkr.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vkr_\w+)", nm))
    assert declared <= exported
    # nothing but the ABI leaks out of the shared object
    leaked = [s for s in re.findall(r" T (\w+)", nm) if not s.startswith("vkr_") and not s.startswith("_")]
    assert leaked == []


def test_struct_layouts_match_the_header(vkr):
    assert C.sizeof(vkr.Image2D) == 24
    assert C.sizeof(vkr.DeferredArgs) == 5 * 24 + 8 + 4 + 80 + 16 + 4  # 4 bytes of padding after light_count
    assert vkr.DeferredArgs.push.offset == 132 and vkr.DeferredArgs.row_offset.offset == 212
    assert C.sizeof(vkr.CullArgs) == 8 + 8 + 8 + 96 + 8 + 16 + 8 * 4
    assert vkr.DRAW_CMD_DTYPE.itemsize == 20  # VkDrawIndexedIndirectCommand


def test_no_gpu_means_loud_failure(vkr):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(vkr.VkrError, match="no HIP device|no CPU fallback"):
        vkr.Context(0)


def test_host_math_matches_restated_glm(vkr, oracle):
    fov = float(np.radians(np.float32(70.0)))
    for aspect in (1.0, 16 / 9, 1000 / 800):
        assert np.array_equal(vkr.perspective(fov, aspect, 0.1, 10000.0), oracle.perspective(fov, aspect, 0.1, 10000.0))
    rng = np.random.default_rng(3)
    for _ in range(20):
        eye = rng.standard_normal(3).astype(np.float32) * 50
        fwd = rng.standard_normal(3).astype(np.float32)
        up = np.array([0, -1, 0], np.float32)
        v = vkr.look_at(eye, eye + fwd, up)
        np.testing.assert_allclose(v, oracle.look_at(eye, eye + fwd, up), rtol=2e-6, atol=2e-5)
        p = vkr.perspective(fov, 1.5, 0.1, 10000.0)
        vp = vkr.mat4_mul(p, v)
        np.testing.assert_allclose(vp, oracle.mat_mul(p, v), rtol=1e-5, atol=1e-4)
        inv = vkr.mat4_inverse(vp)
        ref = oracle.inverse(vp)
        # inv(VP)*VP = I to what a binary32 cofactor inverse gives for this badly scaled matrix (cond ~1e5..1e6)
        prod = inv.astype(np.float64).reshape(4, 4).T @ vp.astype(np.float64).reshape(4, 4).T
        assert np.abs(prod - np.eye(4)).max() < 5e-2
        np.testing.assert_allclose(inv, ref, rtol=0, atol=1e-2 * np.abs(ref).max())
        assert np.array_equal(vkr.frustum_planes(vp), oracle.frustum_planes(vp))
    with pytest.raises(vkr.VkrError, match="singular"):
        vkr.mat4_inverse(np.zeros(16, np.float32))


def test_frustum_planes_classify_clip_space_points(vkr, synth):
    cam = synth.reference_camera(1920, 1080)
    vp = cam["vp"].astype(np.float64).reshape(4, 4).T
    planes = cam["planes"].astype(np.float64)
    rng = np.random.default_rng(5)
    pts = cam["pos"].astype(np.float64) + rng.standard_normal((4000, 3)) * 300
    clip = (vp @ np.c_[pts, np.ones(len(pts))].T).T
    inside_clip = (np.abs(clip[:, 0]) <= clip[:, 3]) & (np.abs(clip[:, 1]) <= clip[:, 3]) & (clip[:, 2] >= 0) & (clip[:, 2] <= clip[:, 3])
    d = pts @ planes[:, :3].T + planes[:, 3]
    margin = np.abs(d).min(1) > 1e-3
    assert np.array_equal(inside_clip[margin], (d >= 0).all(1)[margin])
    assert 0.02 < inside_clip.mean() < 0.6


def test_synth_is_deterministic(synth):
    a = synth.deferred_case(48, 20, 7, config_index=1)
    b = synth.deferred_case(48, 20, 7, config_index=1)
    for k in ("albedo", "material", "normal", "depth"):
        assert np.array_equal(a["gbuf"][k], b["gbuf"][k])
    assert np.array_equal(a["lights"], b["lights"])
    c = synth.deferred_case(48, 20, 7, config_index=2)
    assert not np.array_equal(a["gbuf"]["albedo"], c["gbuf"]["albedo"])
    # pinned stream: splitmix64 counter mode
    assert synth.u64_stream(1, 2, 3).tolist() == synth.u64_stream(1, 2, 3).tolist()
    digest = hashlib.sha256(synth.u64_stream(0x5EED0001, 1, 16).tobytes()).hexdigest()
    assert digest == hashlib.sha256(synth.u64_stream(0x5EED0001, 1, 32)[:16].tobytes()).hexdigest()  # word i depends on i only
    s = synth.cull_case(5000, 64)
    assert s["sizes"].sum() == 5000 and s["mesh_base"][0] == 0
    assert np.array_equal(np.repeat(np.arange(64, dtype=np.uint32), s["sizes"]), s["mesh_id"])
    n = s["gbuf"]["normal"] if "gbuf" in s else None
    assert n is None


def test_synthetic_gbuffer_statistics(synth, oracle):
    case = synth.deferred_case(160, 90, 13)
    g = case["gbuf"]
    assert g["depth"].min() >= 0.9 and g["depth"].max() <= 0.9995
    assert g["albedo"][..., 3].min() == 255 and g["normal"][..., 3].min() == 1.0
    nlen = np.linalg.norm(g["normal"][..., :3].astype(np.float64), axis=-1)
    assert 0.45 < nlen.min() and nlen.max() < 2.05
    # ~10 % of the pixels face away from the camera (early-out path)
    P = synth.unproject(case["camera"]["inv_vp"], 160, 90, g["depth"])
    V = case["camera"]["pos"].astype(np.float64) - P
    facing = ((-g["normal"][..., :3].astype(np.float64)) * V).sum(-1) > 0
    assert 0.85 < facing.mean() < 0.95
    assert (g["material"][..., 1] < 21).mean() > 0.03  # roughness below the alpha floor is exercised


def test_partition_helpers(vkr):
    dist = __import__("importlib").import_module("vk-renderer_amd.dist")
    for h, w in ((2160, 8), (4320, 8), (1080, 7), (5, 8), (1, 3)):
        rows = [dist.band_rows(h, w, r) for r in range(w)]
        assert rows[0][0] == 0 and rows[-1][1] == h
        assert all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
        sizes = [b - a for a, b in rows]
        assert max(sizes) - min(sizes) <= 1
    assert list(dist.frame_slices(64, 8, 3)) == list(range(3, 64, 8))
    base = np.array([0, 0, 10, 10, 500, 900, 1000], np.uint32)  # zero-size meshes included
    for world in (1, 2, 3, 8):
        cuts = dist.mesh_range_cuts(base, 1200, world)
        assert cuts[0][0] == 0 and cuts[0][2] == 0 and cuts[-1][1] == 7 and cuts[-1][3] == 1200
        for a, b in zip(cuts, cuts[1:]):
            assert a[1] == b[0] and a[3] == b[2]
        for m0, m1, i0, i1 in cuts:
            assert m0 <= m1 and i0 <= i1
            if m0 < m1:
                assert i0 == base[m0]


GLOO_WORKER = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
import oracle
synth = importlib.import_module("vk-renderer_amd.synth")
vdist = importlib.import_module("vk-renderer_amd.dist")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
# screen bands: each rank shades its rows with the stand-in compute, gathered frame == single-process frame
case = synth.deferred_case(40, 23, 5, config_index=3)
def band(r0, r1):
    return oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", rows=(r0, r1))["rgba8"][r0:r1]
frame = vdist.sharded_deferred(23, 2, rank, band, vdist.torch_all_gather_numpy)
full = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32")["rgba8"]
assert np.array_equal(frame, full), "band gather differs"
# instance ranges cut on mesh boundaries
scene = synth.cull_case(3000, 37, config_index=4)
def rng(m0, m1, i0, i1):
    res = oracle.cull(scene["aabb_min"][i0:i1], scene["aabb_max"][i0:i1], scene["planes"], scene["mesh_base"][m0:m1] - np.uint32(i0),
                      scene["cmd_template"][m0:m1]) if m1 > m0 else {{"cmds": np.zeros(0, oracle.DRAW_CMD_DTYPE), "visible": np.zeros(0, np.uint32)}}
    cm, vis = res["cmds"].copy(), res["visible"].copy()
    if m1 > m0:
        # the oracle works on local ids/indices: globalise like first_mesh / first_instance do on the GPU
        sub = scene["aabb_min"][i0:i1].copy()
        sub[:, 3] = (sub[:, 3].view(np.uint32) - np.uint32(m0)).view(np.float32)
        res = oracle.cull(sub, scene["aabb_max"][i0:i1], scene["planes"], scene["mesh_base"][m0:m1] - np.uint32(i0), scene["cmd_template"][m0:m1])
        cm, vis = res["cmds"].copy(), res["visible"].copy()
        cm["firstInstance"] += np.uint32(i0)
        vis = np.where(vis == 0xFFFFFFFF, vis, vis + np.uint32(i0))
    return cm, vis
cmds, visible = vdist.sharded_cull(scene["mesh_base"], 3000, 2, rank, rng, vdist.torch_all_gather_numpy)
ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
assert cmds.tobytes() == ref["cmds"].tobytes(), "sharded command buffer differs"
assert np.array_equal(visible, ref["visible"]), "sharded visible list differs"
# ---- the double-buffered gather-to-rank-0 pipelines (what bench.py --gpus N times), oracle injected as compute
class Checked(vdist.HostTransport):
    """Fails the run if a send buffer is handed to a fill while a gather that read it has not been waited for."""
    def __init__(self):
        super().__init__(device="cpu")
        self.reading = {{}}  # data_ptr -> work not yet waited for
        self.gathers = self.waits = 0
    def gather_async(self, send, recv_list):
        assert send.data_ptr() not in self.reading, "gather issued on a buffer whose previous gather is still pending"
        w = super().gather_async(send, recv_list)
        self.reading[send.data_ptr()] = w
        self.gathers += 1
        return w
    def wait(self, work):
        super().wait(work)
        if work is not None:
            self.waits += 1
            for k in [k for k, v in self.reading.items() if v is work]:
                del self.reading[k]
tr = Checked()
W, H = 40, 23  # 23 rows over 2 ranks: bands of 12 and 11 rows
pitch = W * 4 + 16
frames = [synth.deferred_case(W, H, 5, config_index=3 + k) for k in range(5)]
state = {{"k": 0}}
def resolve_band(buf, r0, r1):
    assert buf.data_ptr() not in tr.reading, "buffer rewritten before its gather was waited for"
    c = frames[state["k"]]
    img = oracle.deferred(c["gbuf"], c["lights"], c["push"], mode="f32", rows=(r0, r1))["rgba8"][r0:r1]
    view = buf.numpy().reshape(-1, pitch)
    view[: r1 - r0, : W * 4] = img.reshape(r1 - r0, W * 4)
bands = vdist.BandShardedResolve(tr, W, H, pitch, resolve_band)
assert (bands.r0, bands.r1) == ((0, 12) if rank == 0 else (12, 23)) and bands.rows_max == 12
for k in range(5):  # five frames through two buffers: every buffer is reused at least once
    state["k"] = k
    bands.step()
    if k in (1, 4):
        got = bands.frame()
        if rank == 0:
            c = frames[k]
            want = oracle.deferred(c["gbuf"], c["lights"], c["push"], mode="f32")["rgba8"]
            assert np.array_equal(got, want), f"pipelined band gather differs at frame {{k}}"
        else:
            assert got is None
assert tr.gathers == 5 and tr.waits == 5 and not tr.reading
# timed_steps drives the same pipeline (warm-up + steps, drained on both sides)
state["k"] = 2
seconds = vdist.timed_steps(tr, bands, 3, 1, lambda: None)
assert seconds > 0 and not tr.reading
got = bands.frame()
if rank == 0:
    assert np.array_equal(got, oracle.deferred(frames[2]["gbuf"], frames[2]["lights"], frames[2]["push"], mode="f32")["rgba8"])
# instance ranges: commands of this rank's meshes into the gathered buffer
def cull_range(buf, m0, m1, i0, i1):
    assert buf.data_ptr() not in tr.reading
    cm, _ = rng(m0, m1, i0, i1)
    raw = np.ascontiguousarray(cm).view(np.uint8).reshape(-1)
    buf.numpy()[: raw.size] = raw
ranges = vdist.RangeShardedCull(tr, scene["mesh_base"], 3000, cull_range)
assert ranges.cuts == vdist.mesh_range_cuts(scene["mesh_base"], 3000, 2)
for _ in range(3):
    ranges.step()
cmd_bytes = ranges.commands()
if rank == 0:
    assert cmd_bytes == ref["cmds"].tobytes(), "pipelined command gather differs"
else:
    assert cmd_bytes is None
assert not tr.reading
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_process_gloo_sharding(tmp_path):
    """world_size 2 on CPU (gloo): the band gather and the mesh-range cull assembly reproduce the
    single-process result, through the all-gather helpers and through the double-buffered gather-to-rank-0
    pipelines that bench.py --gpus N times (uneven bands, buffers reused only after their gather was waited
    for).  The per-rank compute is the oracle standing in for the HIP entry points."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER.format(root=ROOT, port=port))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out}"
        assert f"rank {r} ok" in out


def test_cull_object_code_has_no_float_fma():
    """The cull kernels must evaluate ((nx*px + ny*py) + nz*pz) + nw with separate multiplies and adds."""
    obj = os.path.join(ROOT, "vk-renderer_amd", "lib", "cull.o")
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not (os.path.exists(obj) and all(os.path.exists(t) for t in tools)):
        pytest.skip("object file or LLVM tools not present")
    tmp = os.environ.get("TMPDIR", "/tmp")
    fat, co = os.path.join(tmp, "vkr_cull_fat.bin"), os.path.join(tmp, "vkr_cull_gfx950.co")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    asm = subprocess.run([tools[2], "-d", co], capture_output=True, text=True, check=True).stdout
    assert "cull_atomic_kernel" in asm and "cull_onepass_kernel" in asm
    assert re.search(r"v_(pk_)?mul_f32", asm) and re.search(r"v_(pk_)?add_f32", asm)
    fused = re.findall(r"\bv_(?:fma|fmac|mad|mac|pk_fma)_(?:f32|legacy_f32|f16)\w*", asm)
    assert fused == [], f"fused multiply-add in cull code: {set(fused)}"


@pytest.mark.parametrize("unit", ["deferred", "gbuffer"])
def test_no_conversion_is_fused_with_arithmetic(unit):
    """A colour is rounded to binary32 and converted to half from there: a multiply-add folded into the conversion
    (v_fma_mix*_f16 / v_mad_mix*_f16 round once, straight to half) gives other bits at half-way points."""
    obj = os.path.join(ROOT, "vk-renderer_amd", "lib", unit + ".o")
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not (os.path.exists(obj) and all(os.path.exists(t) for t in tools)):
        pytest.skip("object file or LLVM tools not present")
    tmp = os.environ.get("TMPDIR", "/tmp")
    fat, co = os.path.join(tmp, f"vkr_{unit}_mix_fat.bin"), os.path.join(tmp, f"vkr_{unit}_mix_gfx950.co")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    asm = subprocess.run([tools[2], "-d", co], capture_output=True, text=True, check=True).stdout
    assert re.search(r"v_cvt_(pk_)?f16_f32", asm)  # the conversions are there, as instructions of their own
    mixed = re.findall(r"\bv_(?:fma|mad)_mix(?:lo|hi)_f16\w*", asm)
    assert mixed == [], f"arithmetic fused with the conversion to half in {unit}: {set(mixed)}"


def test_resolve_kernels_use_no_scratch_memory():
    """The resolve kernels that production paths launch (whole-table, list building and list shading, every output format) must keep all
    their state in registers and LDS: a spilled register is HBM traffic of the order of the output itself (an earlier
    build wrote 61 MB per 4K frame instead of 33 MB because of four spilled registers)."""
    obj = os.path.join(ROOT, "vk-renderer_amd", "lib", "deferred.o")
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not (os.path.exists(obj) and all(os.path.exists(t) for t in tools)):
        pytest.skip("object file or LLVM tools not present")
    tmp = os.environ.get("TMPDIR", "/tmp")
    fat, co = os.path.join(tmp, "vkr_deferred_fat.bin"), os.path.join(tmp, "vkr_deferred_gfx950.co")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True, check=True).stdout
    kernels = {}
    for block in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1)) for k in ("private_segment_fixed_size", "vgpr_spill_count", "vgpr_count")}
    checked = 0
    for name, k in kernels.items():
        flat = re.search(r"deferred_kernelILi(\d)ELb([01])ELi(\d)EE", name)
        pixel = re.search(r"deferred_pixel_kernelILi(\d)ELb([01])EE", name)
        production = (flat and flat.group(2) == "0") or (pixel and pixel.group(2) == "0") or "deferred_lists_kernel" in name or "tile_lists_kernel" in name or "light_bins_kernel" in name
        if not production:
            continue  # the whole-table loop with the per-pixel cutoff is a diagnostic, not a production path
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (name, k)
        assert k["vgpr_count"] <= 128, (name, k)  # four waves per SIMD
        checked += 1
    assert checked == 8 + 4 + 4 + 1  # four output formats each of: whole-table kernel (four pixels per lane, and one), list shading (strips and deep tiles in one kernel), list building; the bins


@pytest.mark.parametrize("unit", ["cull", "scene", "gbuffer"])
def test_other_kernels_use_no_scratch_memory(unit):
    obj = os.path.join(ROOT, "vk-renderer_amd", "lib", unit + ".o")
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not (os.path.exists(obj) and all(os.path.exists(t) for t in tools)):
        pytest.skip("object file or LLVM tools not present")
    tmp = os.environ.get("TMPDIR", "/tmp")
    fat, co = os.path.join(tmp, f"vkr_{unit}_fat.bin"), os.path.join(tmp, f"vkr_{unit}_gfx950.co")
    subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj], check=True, capture_output=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}"],
                   check=True, capture_output=True)
    notes = subprocess.run([tools[2], "--notes", co], capture_output=True, text=True, check=True).stdout
    blocks = notes.split("- .agpr_count")[1:]
    assert blocks
    for block in blocks:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)) == 0, name
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", block).group(1)) == 0, name


def _fma32(a, b, c):
    """float(a * b + c) for binary32 operands: the product of two binary32 numbers is exact in binary64, the sum is
    rounded to binary64 and then to binary32 (a double rounding that can differ from a fused multiply-add by half a
    binary32 ulp in one case in ~2^29: far inside the margin under test)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def test_the_per_pixel_list_test_is_one_sided(synth):
    """The resolve's per-pixel light lists (deferred_walk_kernel) leave a light off a pixel's list when
    l.n' - p.n' + margin < 0, with n' the shading normal times sqrt(1e7) and margin = 2^-20 * sqrt(1e7) * (|p|_1 + max |l|_1)
    + 1e-15 - three multiply-adds on the light's position.  A light may only be left off if the loop that evaluates every
    light would give it weights of exactly zero, i.e. if that loop's own l.n' - the multiply-add chain over the rounded
    light vector l - p - is negative.  Restated here in binary32 and checked on random pairs, on lights within a few ulps
    of tangent planes, and on positions of very different magnitudes: never a light dropped that the loop would keep."""
    rng = np.random.default_rng(7)
    S = np.float32(3162.27766)
    n_pairs = 2_000_000
    scale = np.float32(10.0) ** rng.integers(-3, 6, n_pairs).astype(np.float32)
    p = (rng.standard_normal((n_pairs, 3)).astype(np.float32) * scale[:, None]).astype(np.float32)
    n = rng.standard_normal((n_pairs, 3)).astype(np.float32)
    n = (n / np.linalg.norm(n, axis=1, keepdims=True)).astype(np.float32)
    ns = (n * S).astype(np.float32)
    # lights: a third anywhere, two thirds on the pixel's tangent plane, displaced along the normal by 0 .. a few 1e-7 of
    # the magnitudes involved, in both directions
    t = np.cross(n, rng.standard_normal((n_pairs, 3))).astype(np.float64)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    lscale = np.float32(10.0) ** rng.integers(-3, 6, n_pairs).astype(np.float32)
    on_plane = p.astype(np.float64) + t * (rng.uniform(0.01, 100.0, n_pairs) * lscale)[:, None]
    mag = np.abs(on_plane).sum(1) + np.abs(p.astype(np.float64)).sum(1)
    nudge = rng.choice([0.0, 1e-8, -1e-8, 1e-7, -1e-7, 3e-7, -3e-7, 1e-6, -1e-6, 3e-6, -3e-6], n_pairs)
    on_plane = on_plane + n.astype(np.float64) * (nudge * mag)[:, None]
    anywhere = rng.standard_normal((n_pairs, 3)) * lscale[:, None]
    l = np.where((np.arange(n_pairs) % 3 == 0)[:, None], anywhere, on_plane).astype(np.float32)
    lmax = np.float32(np.abs(l).sum(1).max())  # the kernel uses the table's largest |l|_1; here every pair also gets its own
    for lm in (lmax, None):
        l1 = np.abs(l).sum(1).astype(np.float32) if lm is None else np.full(n_pairs, lm, np.float32)
        p1 = ((np.abs(p[:, 0]) + np.abs(p[:, 1])) + np.abs(p[:, 2])).astype(np.float32)
        pn = _fma32(p[:, 2], ns[:, 2], _fma32(p[:, 1], ns[:, 1], (p[:, 0] * ns[:, 0]).astype(np.float32)))
        margin = _fma32(np.full(n_pairs, np.float32(2.0 ** -20) * S, np.float32), (p1 + l1).astype(np.float32), np.full(n_pairs, 1e-15, np.float32))
        cst = (margin - pn).astype(np.float32)
        fast = _fma32(l[:, 2], ns[:, 2], _fma32(l[:, 1], ns[:, 1], _fma32(l[:, 0], ns[:, 0], cst)))
        w = (l - p).astype(np.float32)
        loop = _fma32(w[:, 2], ns[:, 2], _fma32(w[:, 1], ns[:, 1], (w[:, 0] * ns[:, 0]).astype(np.float32)))
        dropped = np.signbit(fast)
        assert dropped.any() and (~dropped).any()
        assert (loop[dropped] < 0).all(), "a light left off a list whose cosine the loop would not find negative"
        # and not by a hair: the loop's value is at least a third of the margin below zero
        assert (loop[dropped] <= -0.3 * margin[dropped]).all()
    # the test is not vacuous near the plane: with each pair's own |l|_1 most lights a clear distance behind are dropped
    behind = loop < -4.0 * margin
    assert dropped[behind].mean() > 0.99
