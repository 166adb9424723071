"""GPU parity of the frustum cull + indirect-command compaction against the CPU oracle, through
the C ABI.  Bar: bit-exact.  Command buffers are compared byte for byte in both modes; the
visible-index lists are compared raw in deterministic mode and after a per-mesh sort in atomic mode
(wave arrival order is not defined)."""
import os

import numpy as np
import pytest

from _util import sorted_segments

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run(vkr, ctx, scene, planes, flags, **kw):
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"], **kw)
    cs.visible.fill_u32(0xFFFFFFFF)
    cs.run(planes, flags)
    ctx.wait_idle()
    return cs.results(), cs


def _compare(res, ref, scene, deterministic):
    assert res["error"] == 0
    assert res["cmds"].tobytes() == ref["cmds"].tobytes(), "indirect command buffer differs"
    counts = ref["cmds"]["instanceCount"]
    if deterministic:
        assert np.array_equal(res["visible"], ref["visible"]), "compacted list differs (deterministic mode)"
    else:
        for a, b in zip(sorted_segments(res["visible"], scene["mesh_base"], counts),
                        sorted_segments(ref["visible"], scene["mesh_base"], counts)):
            assert np.array_equal(a, b)
        # slots past each mesh's count were not written
        written = np.zeros(res["visible"].shape[0], bool)
        for b, c in zip(scene["mesh_base"], counts):
            written[int(b):int(b) + int(c)] = True
        assert np.all(res["visible"][~written] == 0xFFFFFFFF)


@pytest.mark.parametrize("deterministic", [False, True])
@pytest.mark.parametrize("n,m", [(4096, 16), (10000, 1024), (1, 1), (63, 2), (64, 1), (65, 3), (255, 7), (256, 1), (257, 300),
                                 (1023, 5), (1024, 5), (1025, 5), (100000, 37)])
def test_parity(vkr, synth, oracle, ctx, n, m, deterministic):
    scene = synth.cull_case(n, m, config_index=n % 17)
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
    res, _ = _run(vkr, ctx, scene, scene["planes"], vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, ref, scene, deterministic)


@pytest.mark.parametrize("n,m", [(10000, 1024), (257, 300), (100000, 37), (1 << 20, 512)])
def test_commands_reset_ahead_of_the_call(vkr, synth, oracle, ctx, n, m):
    """VKR_CULL_CMDS_PREZEROED: the caller has put the commands into their starting state with
    vkr_prep_indirect_reset (a frame loop does that behind its draws) and the atomic cull skips its own launch for
    it.  Three frames with moving planes, each reset -> cull: the oracle's bytes every time; the reset alone leaves
    the templates with zero counts; the deterministic mode takes the flag and needs no reset."""
    scene = synth.cull_case(n, m, config_index=n % 13)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    cs.cmd_out.fill_u32(0xDEADBEEF, 0, cs.m * 20 // 4 * 4)
    ctx.prep_indirect_reset(cs.args(scene["planes"], 0))
    ctx.wait_idle()
    start = cs.results()["cmds"]
    want = scene["cmd_template"].copy()
    want["instanceCount"] = 0
    want["firstInstance"] = scene["mesh_base"]
    assert start.tobytes() == want.tobytes()
    for frame in range(3):
        planes = np.asarray(scene["planes"], np.float32).copy()
        planes[:, 3] += np.float32(3.0 * frame)  # every plane moves outwards: more instances pass
        ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], planes, scene["mesh_base"], scene["cmd_template"])
        cs.visible.fill_u32(0xFFFFFFFF)
        ctx.prep_indirect(cs.args(planes, vkr.CULL_CMDS_PREZEROED))
        ctx.wait_idle()
        _compare(cs.results(), ref, scene, False)
        ctx.prep_indirect_reset(cs.args(planes, 0))  # behind the frame's draws
    cs.cmd_out.fill_u32(0xDEADBEEF, 0, cs.m * 20 // 4 * 4)
    cs.visible.fill_u32(0xFFFFFFFF)
    ctx.prep_indirect(cs.args(scene["planes"], vkr.CULL_DETERMINISTIC | vkr.CULL_CMDS_PREZEROED))
    ctx.wait_idle()
    _compare(cs.results(), oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"]), scene, True)


@pytest.mark.parametrize("deterministic", [False, True])
def test_golden_fixture(vkr, ctx, deterministic):
    z = np.load(os.path.join(GOLDEN, "cull_4096_M16.npz"))
    scene = {k: z[k] for k in ("aabb_min", "aabb_max", "mesh_base", "cmd_template")}
    res, _ = _run(vkr, ctx, scene, z["planes"], vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, {"cmds": z["cmds"], "visible": z["visible"]}, scene, deterministic)


def test_adversarial_near_plane(vkr, synth, oracle, ctx):
    """AABBs whose deciding corner is within 2 ulp of a plane: any FMA contraction or re-association on
    the GPU flips verdicts."""
    cam = synth.reference_camera(3840, 2160)
    scene = synth.make_adversarial_cull(65536, synth.SEED_CULL + 99, cam, 16)
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], cam["planes"], scene["mesh_base"], scene["cmd_template"])
    z = np.load(os.path.join(GOLDEN, "cull_adversarial_65536.npz"))
    assert z["cmds"].tobytes() == ref["cmds"].tobytes()  # the oracle still reproduces its committed answer
    frac = ref["mask"].mean()
    assert 0.2 < frac < 0.8
    for flags in (0, vkr.CULL_DETERMINISTIC):
        res, _ = _run(vkr, ctx, scene, cam["planes"], flags)
        _compare(res, ref, scene, bool(flags))


@pytest.mark.parametrize("deterministic", [False, True])
def test_full_4m(vkr, synth, oracle, ctx, deterministic):
    """BASELINE size: 2^22 instances, 1024 meshes, complete comparison with the oracle."""
    n = 1 << 22
    scene = synth.cull_case(n, 1024, config_index=2)
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
    frac = ref["mask"].mean()
    assert 0.05 < frac < 0.5
    res, cs = _run(vkr, ctx, scene, scene["planes"], vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, ref, scene, deterministic)
    assert int(res["cmds"]["instanceCount"].sum()) == int(ref["mask"].sum())
    if not deterministic:
        # idempotence: a second run over the same inputs gives the same commands and the same sets
        cs.visible.fill_u32(0xFFFFFFFF)
        cs.run(scene["planes"], 0)
        ctx.wait_idle()
        _compare(cs.results(), ref, scene, False)


def _with_segments(scene, vkr, sizes):
    """The same boxes under a hand-made segment table: mesh k owns sizes[k] consecutive instances."""
    sizes = np.asarray(sizes, np.int64)
    n = scene["aabb_min"].shape[0]
    assert int(sizes.sum()) == n
    base = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint32)
    mesh_of = np.repeat(np.arange(sizes.shape[0], dtype=np.uint32), sizes)
    mn = scene["aabb_min"].copy()
    mn[:, 3] = mesh_of.view(np.float32)
    tmpl = np.zeros(sizes.shape[0], vkr.DRAW_CMD_DTYPE)
    tmpl["indexCount"] = 3 * (1 + np.arange(sizes.shape[0]))
    tmpl["firstIndex"] = 7 * np.arange(sizes.shape[0])
    tmpl["vertexOffset"] = 11 * np.arange(sizes.shape[0])
    return {"aabb_min": mn, "aabb_max": scene["aabb_max"], "mesh_base": base, "cmd_template": tmpl}


@pytest.mark.parametrize("sizes", [
    [0, 0, 5000, 0, 0],                                  # empty meshes first and last
    [1024, 0, 1024, 0, 0, 2048, 904],                    # boundaries exactly on workgroup edges, empties on them
    [1, 1023, 1, 0, 1024, 2047, 1, 903],                 # boundaries one off the edges
    [0] * 300 + [5000] + [0] * 300,                      # more empty meshes than threads in a workgroup
    [1] * 2500 + [2500],                                 # more meshes than instances per workgroup
    [5000],
    [2048, 2952],                                        # a mesh that ends exactly where a block ends
    [1000, 24, 3976],                                    # a mesh that begins and ends inside one block
])
@pytest.mark.parametrize("deterministic", [False, True])
def test_segment_table_edges(vkr, synth, oracle, ctx, sizes, deterministic):
    scene = _with_segments(synth.cull_case(5000, 4, config_index=3), vkr, sizes)
    planes = synth.cull_case(16, 1)["planes"]
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], planes, scene["mesh_base"], scene["cmd_template"])
    assert 0 < int(ref["mask"].sum()) < 5000
    res, _ = _run(vkr, ctx, scene, planes, vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, ref, scene, deterministic)
    # and with every instance visible
    everything = np.zeros((6, 4), np.float32)
    everything[:, 3] = 1.0
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], everything, scene["mesh_base"], scene["cmd_template"])
    assert int(ref["mask"].sum()) == 5000
    res, _ = _run(vkr, ctx, scene, everything, vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, ref, scene, deterministic)


@pytest.mark.parametrize("n,m", [(8192 * 1024 + 1, 777), (9_000_037, 777), (3_000_000, 3), (3_000_000, 1)])
def test_deterministic_many_blocks_per_workgroup_and_meshes_of_thousands_of_blocks(vkr, synth, oracle, ctx, n, m):
    """More blocks of 1024 instances than the resident grid has workgroups (each workgroup walks several), and
    meshes that span thousands of blocks (the look-back runs over many 64-block steps).  Complete comparison,
    three calls in a row on the same state (sequence numbers, not clearing, separate the calls)."""
    scene = synth.cull_case(n, m, config_index=4)
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
    for _ in range(3):
        res, _ = _run(vkr, ctx, scene, scene["planes"], vkr.CULL_DETERMINISTIC)
        _compare(res, ref, scene, True)


def test_instance_range_sharding(vkr, synth, oracle, ctx):
    """Two calls over disjoint mesh ranges (cut at a mesh boundary) assemble into the single-call
    result: mesh ids stay global (first_mesh), indices are global (first_instance)."""
    n, m = 50000, 64
    scene = synth.cull_case(n, m, config_index=5)
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
    cut_mesh = 29
    cut = int(scene["mesh_base"][cut_mesh])
    cmds = np.zeros(m, vkr.DRAW_CMD_DTYPE)
    visible = np.full(n, 0xFFFFFFFF, np.uint32)
    for (m0, m1, i0, i1) in ((0, cut_mesh, 0, cut), (cut_mesh, m, cut, n)):
        part = {"aabb_min": scene["aabb_min"][i0:i1], "aabb_max": scene["aabb_max"][i0:i1],
                "mesh_base": scene["mesh_base"][m0:m1] - np.uint32(i0), "cmd_template": scene["cmd_template"][m0:m1]}
        res, _ = _run(vkr, ctx, part, scene["planes"], vkr.CULL_DETERMINISTIC, first_mesh=m0, first_instance=i0)
        assert res["error"] == 0
        cmds[m0:m1] = res["cmds"]
        visible[i0:i1] = res["visible"]
    assert cmds.tobytes() == ref["cmds"].tobytes()
    assert np.array_equal(visible, ref["visible"])


def test_bad_mesh_id_sets_error_flag(vkr, synth, ctx):
    scene = synth.cull_case(2000, 8, config_index=1)
    bad = scene["aabb_min"].copy()
    bad[:, 3] = np.full(2000, 8, np.uint32).view(np.float32)  # mesh 8 does not exist
    scene["aabb_min"] = bad
    # a frustum that contains everything, so the bad instances are reached
    planes = np.zeros((6, 4), np.float32)
    planes[:, 3] = 1.0
    for flags in (0, vkr.CULL_DETERMINISTIC):
        res, _ = _run(vkr, ctx, scene, planes, flags)
        assert res["error"] == 1
        assert int(res["cmds"]["instanceCount"].sum()) == 0


def test_nan_box_is_culled_and_everything_visible_case(vkr, synth, oracle, ctx):
    scene = synth.cull_case(777, 4, config_index=8)
    scene["aabb_min"] = scene["aabb_min"].copy()
    scene["aabb_min"][5, 0] = np.nan
    planes = np.zeros((6, 4), np.float32)
    planes[:, 3] = 1.0  # d = 1 >= 0 for every finite box
    planes[0, 0] = 1e-30
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], planes, scene["mesh_base"], scene["cmd_template"])
    assert int(ref["mask"].sum()) == 777 - 0  # plane 0 picks max.x (finite): the NaN min is never read
    planes[0, 0] = -1e-30  # now the p-vertex takes min.x = NaN -> culled
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], planes, scene["mesh_base"], scene["cmd_template"])
    assert int(ref["mask"].sum()) == 776
    res, _ = _run(vkr, ctx, scene, planes, vkr.CULL_DETERMINISTIC)
    _compare(res, ref, scene, True)


def test_argument_errors(vkr, synth, ctx):
    scene = synth.cull_case(100, 4)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    a = cs.args(scene["planes"], 0)
    a.visible = None
    with pytest.raises(vkr.VkrError, match="null"):
        ctx.prep_indirect(a)
    a = cs.args(scene["planes"], 0x40)
    with pytest.raises(vkr.VkrError, match="flags"):
        ctx.prep_indirect(a)
    # zero instances / zero meshes are legal no-ops
    a = cs.args(scene["planes"], 0)
    a.instance_count = 0
    ctx.prep_indirect(a)
    ctx.wait_idle()
    assert int(cs.cmd_out.download(vkr.DRAW_CMD_DTYPE, 4)["instanceCount"].sum()) == 0


@pytest.mark.parametrize("deterministic", [False, True])
def test_subnormal_distances_keep_their_sign(vkr, synth, oracle, ctx, deterministic):
    """Plane distances that are subnormal numbers: a device that flushed them to zero would call a box visible
    (-0 >= 0) that the binary32 evaluation on the CPU culls.  Boxes a few 1e-39 either side of a plane through
    the origin, products and sums all in the subnormal range."""
    n, m = 4096, 4
    scene = synth.cull_case(n, m, config_index=12)
    rng = np.random.default_rng(12)
    mn = scene["aabb_min"].copy()
    mx = scene["aabb_max"].copy()
    tiny = (rng.integers(-2000, 2000, (n, 3)).astype(np.float64) * 1e-42).astype(np.float32)  # multiples of the smallest subnormal
    mn[:, :3] = tiny
    mx[:, :3] = tiny + np.float32(3e-42)
    scene["aabb_min"], scene["aabb_max"] = mn, mx
    planes = np.zeros((6, 4), np.float32)
    planes[:, 3] = 1.0
    planes[0] = [1.0, 0.0, 0.0, -1e-39]    # x >= 1e-39 at the p-vertex
    planes[1] = [0.0, -0.5, 0.0, 2e-40]    # -0.5 y + 2e-40 >= 0
    planes[2] = [0.25, 0.25, 0.25, 0.0]
    ref = oracle.cull(mn, mx, planes, scene["mesh_base"], scene["cmd_template"])
    assert 0.02 < ref["mask"].mean() < 0.98  # the planes really split the set
    res, _ = _run(vkr, ctx, scene, planes, vkr.CULL_DETERMINISTIC if deterministic else 0)
    _compare(res, ref, scene, deterministic)
