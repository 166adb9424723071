"""The entry point that allocates nothing, vkr_prep_indirect_noalloc, against the existing deterministic entry
point (vkr_prep_indirect with VKR_CULL_DETERMINISTIC, whose state lives in context-owned hipMalloc'd memory),
through the C ABI.  Bar: bit-exact - command buffer byte for byte, compacted list word for word (slots past
each mesh's count included: neither call writes them), error flag.  Two calls in a row share the library's
one static array per GPU, the second on a smaller input, so anything the larger call left behind would show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(vkr, ctx, scene, planes, **kw):
    """(results of the existing entry point, results of the allocation-free one) on the same device inputs."""
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"], **kw)
    out = []
    for noalloc in (False, True):
        cs.visible.fill_u32(0xFFFFFFFF)
        cs.cmd_out.fill_u32(0xDEADBEEF, 0, cs.m * 20 // 4 * 4)
        cs.error_flag.fill_u32(0)
        if noalloc:
            ctx.prep_indirect_noalloc(cs.args(planes, 0))  # VKR_CULL_DETERMINISTIC is implied
        else:
            ctx.prep_indirect(cs.args(planes, vkr.CULL_DETERMINISTIC))
        ctx.wait_idle()
        out.append(cs.results())
    return out


def _same(a, b):
    assert a["error"] == b["error"] == 0
    assert a["cmds"].tobytes() == b["cmds"].tobytes(), "indirect command buffer differs"
    assert np.array_equal(a["visible"], b["visible"]), "compacted list differs"


def test_two_calls_in_a_row_second_smaller(vkr, synth, oracle, ctx):
    big = synth.cull_case(100000, 37, config_index=6)
    small = synth.cull_case(5000, 9, config_index=7)
    for scene in (big, small):
        want, got = _both(vkr, ctx, scene, scene["planes"])
        _same(want, got)
        ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
        assert got["cmds"].tobytes() == ref["cmds"].tobytes()
        assert np.array_equal(got["visible"], ref["visible"])
        assert 0 < int(ref["mask"].sum()) < scene["aabb_min"].shape[0]


def _free_device_bytes():
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_largest_accepted_size_then_a_ragged_one(vkr, synth, oracle, ctx):
    n = vkr.NOALLOC_MAX_INSTANCES
    scene = synth.cull_case(n, 1024, config_index=2)
    want, got = _both(vkr, ctx, scene, scene["planes"])
    _same(want, got)
    assert 0.05 < int(got["cmds"]["instanceCount"].sum()) / n < 0.5
    ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"])
    assert got["cmds"].tobytes() == ref["cmds"].tobytes() and np.array_equal(got["visible"], ref["visible"])
    ragged = synth.cull_case(1025, 5, config_index=3)
    want, got = _both(vkr, ctx, ragged, ragged["planes"])
    _same(want, got)


def test_calls_leave_the_free_device_memory_as_it_was(vkr, synth, ctx):
    """hipMemGetInfo before and after calls of every accepted size class, the largest included: not a byte taken
    ."""
    scenes = [synth.cull_case(n, m, config_index=2) for n, m in ((vkr.NOALLOC_MAX_INSTANCES, 1024), (70000, 30), (1, 1))]
    sets = [vkr.CullScene(ctx, sc["aabb_min"], sc["aabb_max"], sc["mesh_base"], sc["cmd_template"]) for sc in scenes]
    sets[1].run(scenes[1]["planes"], 0)  # (the atomic mode: loads nothing the calls below could be charged with)
    ctx.wait_idle()
    before = _free_device_bytes()
    for _ in range(3):
        for cs, sc in zip(sets, scenes):
            ctx.prep_indirect_noalloc(cs.args(sc["planes"], 0))
    ctx.wait_idle()
    assert _free_device_bytes() == before
    block = ctx.buffer(64 << 20)
    assert _free_device_bytes() <= before - (64 << 20)  # the measurement does see an allocation when there is one
    block.destroy()


def test_calls_from_two_contexts_without_a_wait_in_between(vkr, synth, oracle):
    """Every call on a GPU uses the one array, whatever its context: the library makes each launch wait for the one
    before, so two contexts (two streams) may issue calls back to back and still get the oracle's lists."""
    c1, c2 = vkr.Context(0), vkr.Context(0)
    work = []
    for ctx, (n, m, seed) in ((c1, (1 << 21, 500, 1)), (c2, (1 << 20, 64, 2)), (c1, (300000, 7, 3)), (c2, (1 << 21, 900, 4))):
        sc = synth.cull_case(n, m, config_index=seed)
        cs = vkr.CullScene(ctx, sc["aabb_min"], sc["aabb_max"], sc["mesh_base"], sc["cmd_template"])
        cs.visible.fill_u32(0xFFFFFFFF)
        work.append((ctx, sc, cs, oracle.cull(sc["aabb_min"], sc["aabb_max"], sc["planes"], sc["mesh_base"], sc["cmd_template"])))
    c1.wait_idle()
    c2.wait_idle()
    for _ in range(4):
        for ctx, sc, cs, _ref in work:
            ctx.prep_indirect_noalloc(cs.args(sc["planes"], 0))
    c1.wait_idle()
    c2.wait_idle()
    for _ctx, _sc, cs, ref in work:
        res = cs.results()
        assert res["error"] == 0 and res["cmds"].tobytes() == ref["cmds"].tobytes() and np.array_equal(res["visible"], ref["visible"])


def test_instance_range_arguments(vkr, synth, ctx):
    """first_mesh / first_instance behave as in the existing entry point."""
    scene = synth.cull_case(50000, 64, config_index=5)
    m0, i0 = 29, int(scene["mesh_base"][29])
    part = {"aabb_min": scene["aabb_min"][i0:], "aabb_max": scene["aabb_max"][i0:],
            "mesh_base": scene["mesh_base"][m0:] - np.uint32(i0), "cmd_template": scene["cmd_template"][m0:]}
    want, got = _both(vkr, ctx, part, scene["planes"], first_mesh=m0, first_instance=i0)
    _same(want, got)


def test_rejects_sizes_beyond_the_static_array(vkr, synth, ctx):
    scene = synth.cull_case(100, 4)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    a = cs.args(scene["planes"], 0)
    a.instance_count = vkr.NOALLOC_MAX_INSTANCES + 1
    with pytest.raises(vkr.VkrError, match="exceed the limit"):
        ctx.prep_indirect_noalloc(a)
    # an unchanged small call still works afterwards
    ctx.prep_indirect_noalloc(cs.args(scene["planes"], 0))
    ctx.wait_idle()
    assert cs.results()["error"] == 0


def test_recorded_into_a_graph_and_replayed(vkr, synth, ctx):
    """A caller under HIP stream capture is one of those this entry point is for.  A recorded launch runs again
    with the arguments it was recorded with, so it cannot count on a fresh sequence number: the library notices
    the capture, records a clear of the array in front of the launch and tags it with a number of its own.  The
    graph is replayed four times over buffers whose boxes change between replays (words left by the replay before
    would give wrong ranks), with an ordinary call on another scene in between, and must give the bytes of the
    existing entry point every time."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p(ctx.stream())
    scene = synth.cull_case(200000, 50, config_index=4)
    other = synth.cull_case(30000, 11, config_index=9)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])

    def boxes(shift):
        """The same segments with the boxes moved round by `shift` places (mesh ids stay where they are)."""
        mn, mx = np.roll(scene["aabb_min"], shift, axis=0).copy(), np.roll(scene["aabb_max"], shift, axis=0).copy()
        mn[:, 3] = scene["aabb_min"][:, 3]
        return mn, mx

    def expected(shift):
        mn, mx = boxes(shift)
        cs.aabb_min.upload(mn)
        cs.aabb_max.upload(mx)
        cs.visible.fill_u32(0xFFFFFFFF)  # slots past each mesh's count are written by neither call
        ctx.prep_indirect(cs.args(scene["planes"], vkr.CULL_DETERMINISTIC))
        ctx.wait_idle()
        return cs.results()

    shifts = (0, 777, 123456, 0)
    wants = [expected(sh) for sh in shifts]
    assert len({w["cmds"].tobytes() for w in wants[:3]}) == 3  # the replays really have different answers

    graph, graph_exec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 1) == 0  # hipStreamCaptureModeThreadLocal
    try:
        ctx.prep_indirect_noalloc(cs.args(scene["planes"], 0))
    finally:
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
    assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, C.c_size_t(0)) == 0
    try:
        for replay, shift in enumerate(shifts):
            mn, mx = boxes(shift)
            cs.aabb_min.upload(mn)
            cs.aabb_max.upload(mx)
            cs.visible.fill_u32(0xFFFFFFFF)
            cs.cmd_out.fill_u32(0xDEADBEEF, 0, cs.m * 20 // 4 * 4)
            ctx.wait_idle()
            assert hip.hipGraphLaunch(graph_exec, stream) == 0
            ctx.wait_idle()
            _same(wants[replay], cs.results())
            if replay == 1:  # an ordinary call in between leaves its own words in the array
                _same(*_both(vkr, ctx, other, other["planes"]))
    finally:
        hip.hipGraphExecDestroy(graph_exec)
        hip.hipGraphDestroy(graph)


def test_stateful_calls_refuse_to_be_recorded(vkr, synth, ctx):
    """The deterministic cull with context-owned state and the radius-cutoff resolve pass a value to their kernels
    that the host advances per call: recorded, they would replay with a stale one, so they say no.  The plain
    resolve and the atomic cull carry no such state and record fine (their replay is checked against a direct call)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    stream = C.c_void_p(ctx.stream())
    scene = synth.cull_case(20000, 8, config_index=3)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    case = synth.deferred_case(64, 36, 13, config_index=0)
    gb = vkr.GBuffer(ctx, 64, 36).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 64, 36)
    plain = vkr.make_deferred_args(gb, out, lights, 13, case["push"])
    push = case["push"].copy()
    push[3] = 50.0
    cutoff = vkr.make_deferred_args(gb, out, lights, 13, push, flags=vkr.DEFERRED_RADIUS_CUTOFF)
    ctx.deferred_resolve(plain)
    cs.visible.fill_u32(0xFFFFFFFF)
    ctx.prep_indirect(cs.args(scene["planes"], 0))
    ctx.wait_idle()
    want_img, want_cull = out.download(np.uint8, 4), cs.results()

    graph, graph_exec = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 1) == 0
    try:
        with pytest.raises(vkr.VkrError, match="cannot be recorded"):
            ctx.prep_indirect(cs.args(scene["planes"], vkr.CULL_DETERMINISTIC))
        with pytest.raises(vkr.VkrError, match="cannot be recorded"):
            ctx.deferred_resolve(cutoff)
        ctx.deferred_resolve(plain)
        ctx.prep_indirect(cs.args(scene["planes"], 0))
    finally:
        assert hip.hipStreamEndCapture(stream, C.byref(graph)) == 0
    assert hip.hipGraphInstantiate(C.byref(graph_exec), graph, None, None, C.c_size_t(0)) == 0
    try:
        for _ in range(2):
            out.buffer.fill_u32(0)
            cs.visible.fill_u32(0xFFFFFFFF)
            cs.cmd_out.fill_u32(0xDEADBEEF, 0, cs.m * 20 // 4 * 4)
            ctx.wait_idle()
            assert hip.hipGraphLaunch(graph_exec, stream) == 0
            ctx.wait_idle()
            assert np.array_equal(out.download(np.uint8, 4), want_img)
            got = cs.results()
            assert got["cmds"].tobytes() == want_cull["cmds"].tobytes()
            counts = got["cmds"]["instanceCount"]
            for b, c in zip(scene["mesh_base"], counts):
                assert np.array_equal(np.sort(got["visible"][int(b):int(b) + int(c)]), np.sort(want_cull["visible"][int(b):int(b) + int(c)]))
    finally:
        hip.hipGraphExecDestroy(graph_exec)
        hip.hipGraphDestroy(graph)
    # outside capture everything works as before
    ctx.deferred_resolve(cutoff)
    ctx.prep_indirect(cs.args(scene["planes"], vkr.CULL_DETERMINISTIC))
    ctx.wait_idle()
