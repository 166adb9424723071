"""The reference's shader text, run: tests/glsl_eval.py evaluates shaders/deferred.comp itself (read from
/root/reference where that exists - this container; the GPU box has no copy) per pixel in binary64, and the CPU
restatement in oracle/ must give the same colours.  The committed fixture tests/golden/shader_text_eval_*.npz holds
the inputs and the result of such a run (tests/golden/make_shader_eval.py), so the comparison also runs where the
reference is absent, and the GPU parity tests compare the kernels with it."""
import importlib
import os

import numpy as np
import pytest

from glsl_eval import GlslError, Interpreter, run_deferred_shader

HERE = os.path.dirname(os.path.abspath(__file__))
SHADER = "/root/reference/shaders/deferred.comp"
FIXTURE = os.path.join(HERE, "golden", "shader_text_eval_24x10_L13.npz")


def _fixture():
    d = np.load(FIXTURE)
    gbuf = {k: d[k] for k in ("albedo", "material", "normal", "depth")}
    return gbuf, d["lights"], d["push"], d["color"]


def test_evaluator_on_known_snippets():
    src = """
    #define N 3
    float sq(float x) { return x * x; }
    float pick(float x) { return x > 0 ? 1 : 0; }
    vec3 f(vec3 a, float t) {
        float s = 0.0;
        for (int i = 0; i < N; i++) { s += sq(a[i]); }
        if (s <= 1e-6) return vec3(0.0);
        vec3 h = normalize(a);
        s *= mix(1.0, 1.0/1.51, t);
        return h * s + vec3(pick(t), 2, max(1 - t, 0));
    }
    """
    it = Interpreter(src, {}, {})
    a = np.array([1.0, 2.0, 2.0])
    got = it.call("f", [a, 0.5])
    k = float(np.float32(1.0) / np.float32(1.51))  # folded in binary32
    want = a / 3.0 * (9.0 * (1.0 * 0.5 + k * 0.5)) + np.array([1.0, 2.0, 0.5])
    assert np.allclose(got, want, rtol=1e-15, atol=0)
    assert np.array_equal(it.call("f", [np.zeros(3), 0.5]), np.zeros(3))
    with pytest.raises(GlslError):
        Interpreter("float g(float x) { while (x > 0) x -= 1; return x; }", {}, {}).call("g", [1.0])


def test_oracle_equals_the_shader_text_on_the_fixture(oracle):
    """binary64 against binary64: the restatement and the text agree to rounding noise on every pixel."""
    gbuf, lights, push, color = _fixture()
    got = oracle.deferred(gbuf, lights, push, mode="f64", want=("f64",))
    assert not got["near_threshold"].any()
    a, b = got["f64"][..., :3], color[..., :3]
    assert np.abs(a - b).max() <= 1e-12 * np.abs(b).max() + 1e-15, float(np.abs(a - b).max())
    assert (color[..., 3] == 1.0).all() and b.max() > 10 * b.min() > 0  # lit and unlit pixels both present
    # the binary32 restatement, the one the CPU baseline times, within the tolerance the kernels are held to
    o32 = oracle.deferred(gbuf, lights, push, mode="f32", want=("f32",))["f32"][..., :3].astype(np.float64)
    assert (np.abs(o32 - b) <= 1e-4 * np.abs(b) + 2e-6).all()


def test_binary32_restatement_is_the_text_evaluated_in_binary32(oracle):
    """Operation ORDER: the fixture also holds the shader's text evaluated with every operation rounded to binary32
    (normalize as x * (1 / sqrt(dot(x, x))), pow as the C library's powf, matrix * vector as the sum of scaled
    columns).  The binary32 restatement - the one whose association the kernels' position reconstruction copies and
    that the CPU baseline times - gives the same BITS on every channel of every pixel."""
    gbuf, lights, push, _ = _fixture()
    want = np.load(FIXTURE)["color_f32"]
    got = oracle.deferred(gbuf, lights, push, mode="f32", want=("f32",))["f32"]
    assert np.array_equal(got[..., :3].view(np.uint32), want[..., :3].view(np.uint32))


@pytest.mark.skipif(not os.path.exists(SHADER), reason="reference sources not present")
def test_binary32_evaluation_of_the_text_on_other_frames(oracle):
    import glsl_eval
    synth = importlib.import_module("vk-renderer_amd.synth")
    text = open(SHADER).read()
    glsl_eval.set_precision(np.float32)
    try:
        for cfg, (w, h) in ((33, (16, 8)), (5, (20, 6))):
            case = synth.deferred_case(w, h, 13, config_index=cfg)
            with np.errstate(all="ignore"):
                mine = run_deferred_shader(text, case["gbuf"], case["lights"], case["push"], w, h).astype(np.float32)
            theirs = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", want=("f32",))["f32"]
            same = (theirs[..., :3].view(np.uint32) == mine[..., :3].view(np.uint32)) | (np.isnan(theirs[..., :3]) & np.isnan(mine[..., :3]))
            assert same.all(), (cfg, float(same.mean()))
    finally:
        glsl_eval.set_precision(np.float64)


@pytest.mark.skipif(not os.path.exists(SHADER), reason="reference sources not present")
def test_fixture_is_what_the_shader_text_gives_today():
    """Re-runs the reference's text on the fixture's inputs: the stored colours are its output, bit for bit; and on
    a second frame (other seed, lights closer in) the restatement agrees with it as well."""
    gbuf, lights, push, color = _fixture()
    text = open(SHADER).read()
    assert np.array_equal(run_deferred_shader(text, gbuf, lights, push, 24, 10), color)
    import glsl_eval
    glsl_eval.set_precision(np.float32)
    try:
        with np.errstate(all="ignore"):
            again32 = run_deferred_shader(text, gbuf, lights, push, 24, 10).astype(np.float32)
    finally:
        glsl_eval.set_precision(np.float64)
    assert np.array_equal(again32.view(np.uint32), np.load(FIXTURE)["color_f32"].view(np.uint32))
    synth = importlib.import_module("vk-renderer_amd.synth")
    oracle = importlib.import_module("oracle")
    case = synth.deferred_case(9, 7, 13, config_index=33)
    mine = run_deferred_shader(text, case["gbuf"], case["lights"], case["push"], 9, 7)
    theirs = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f64", want=("f64",))
    ok = theirs["near_threshold"] == 0  # a light exactly at the 1e-6 early-out may fall either way
    a, b = theirs["f64"][..., :3][ok], mine[..., :3][ok]
    assert ok.mean() > 0.9 and np.abs(a - b).max() <= 1e-11 * np.abs(b).max() + 1e-15


HARD = ("shader_text_mirror_peak_64x8_L13.npz", "shader_text_near_light_32x8_L13.npz")


@pytest.mark.parametrize("name", HARD)
def test_hard_frames_of_the_shader_text(oracle, name):
    """tests/golden/make_shader_eval_hard.py: a mirror peak at the floor of alpha, and lights a fraction of a unit
    from surfaces 320-384 units out, both with positions that reconstruct exactly.  The binary64 restatement agrees
    with the text's binary64 colours to rounding noise, the binary32 restatement with the text's binary32 colours
    bit for bit; the mixed evaluation the kernels are compared with (binary64 lighting on binary32 positions) is
    the binary64 one here.  How far the LITERAL binary32 run is from binary64 on these frames is asserted too: on
    the mirror peak it is off by up to 0.8 % on half of the channels (the GGX denominator cancels), which is why the
    kernels do not evaluate that term the way the text spells it."""
    z = np.load(os.path.join(HERE, "golden", name))
    gbuf = {k: z[k] for k in ("albedo", "material", "normal", "depth")}
    h, w = z["depth"].shape
    want = z["color"][..., :3]
    o64 = oracle.deferred(gbuf, z["lights"], z["push"], mode="f64", want=("f64",))
    assert not o64["near_threshold"].any()
    assert np.abs(o64["f64"][..., :3] - want).max() <= 1e-11 * np.abs(want).max()
    mix = oracle.deferred(gbuf, z["lights"], z["push"], mode="mix", want=("f64",))["f64"][..., :3]
    assert np.abs(mix - want).max() <= 1e-11 * np.abs(want).max()
    o32 = oracle.deferred(gbuf, z["lights"], z["push"], mode="f32", want=("f32",))["f32"][..., :3]
    assert np.array_equal(o32.view(np.uint32), z["color_f32"][..., :3].view(np.uint32))
    rel = np.abs(z["color_f32"][..., :3].astype(np.float64) - want) / (np.abs(want) + 2e-2)
    if "mirror" in name:
        assert 5e-3 < rel.max() < 1e-2 and 0.4 < (rel > 1e-4).mean() < 0.7
        assert want.max() > 100  # the peak itself is in the frame
    else:
        assert rel.max() < 1e-5
        lights = z["lights"][:10, 0:3].astype(np.float64)
        x = ((np.arange(w) + 0.5) / w * 2 - 1) * float(z["push"][4])
        y = ((np.arange(h) + 0.5) / h * 2 - 1) * float(z["push"][9])
        P = np.stack([np.broadcast_to(x[None, :], (h, w)), np.broadcast_to(y[:, None], (h, w)), z["depth"].astype(np.float64) * float(z["push"][14])], -1)
        d = np.sqrt(((P[:, :, None, :] - lights[None, None]) ** 2).sum(-1)).min((0, 1))
        assert (d <= 0.5001).all() and d.min() < 0.06 and P[..., 2].min() >= 320.0  # what the frame is for
    if os.path.exists(SHADER):
        assert np.array_equal(run_deferred_shader(open(SHADER).read(), gbuf, z["lights"], z["push"], w, h), z["color"])


VERT, FRAG = "/root/reference/shaders/offscreen.vert", "/root/reference/shaders/offscreen.frag"


@pytest.mark.skipif(not (os.path.exists(VERT) and os.path.exists(FRAG)), reason="reference sources not present")
def test_surface_encode_oracle_equals_the_offscreen_shaders_text(oracle):
    """The G-buffer pass's two shaders, their own text, per pixel (vertex stage with the pixel's attributes, then the
    fragment stage) against oracle_gbuffer_encode_surfaces: the attachment values agree to the last step of their
    formats (the text runs in binary64, the restatement in binary32, so a value on a rounding boundary may land
    one step apart)."""
    from glsl_eval import run_offscreen_shaders
    rng = np.random.default_rng(41)
    h, w, K = 6, 16, 5
    normal_obj = rng.normal(size=(h, w, 4)).astype(np.float32)
    tangent_obj = rng.normal(size=(h, w, 4)).astype(np.float32)
    index = rng.integers(0, K, (h, w)).astype(np.uint32)
    normal_obj[..., 3] = index.view(np.float32)
    matrices = (rng.normal(size=(K, 16)) * np.exp(rng.uniform(-2, 2, (K, 1)))).astype(np.float32)
    albedo = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    specular = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    normal_texel = rng.uniform(0, 1, (h, w, 4)).astype(np.float32)
    text = run_offscreen_shaders(open(VERT).read(), open(FRAG).read(), normal_obj, tangent_obj, matrices, index, albedo, specular, normal_texel)
    mine = oracle.gbuffer_encode_surfaces(albedo, specular, normal_texel, normal_obj, tangent_obj, np.zeros((h, w), np.float32), matrices)
    # rgba16f normal: within one half-precision step; w = 1
    n_text, n_mine = text["outNormal"], mine["normal"].astype(np.float64)
    step = np.maximum(np.abs(n_text), 2.0 ** -14) * 2.0 ** -10
    assert (np.abs(n_text - n_mine) <= step).all() and (n_mine[..., 3] == 1.0).all()
    # rgba8 colour and material: within one step of 1/255
    for name, key in (("outColor", "albedo"), ("outSpecular", "material")):
        assert np.abs(np.rint(np.clip(text[name], 0, 1) * 255) - mine[key].astype(np.float64)).max() <= 1, name


@pytest.mark.skipif(not os.path.exists(SHADER), reason="reference sources not present")
def test_shader_text_on_the_13_light_golden_frame(oracle):
    """The committed 64 x 36 frame with the reference's own light count (tests/golden/deferred_64x36_L13.npz, whose
    stored outputs are the oracle's): the shader's text against the binary64 restatement on all 2304 pixels, and
    against the stored mixed-precision colours where the position's last bit does not matter (1e-4)."""
    z = np.load(os.path.join(HERE, "golden", "deferred_64x36_L13.npz"))
    gbuf = {k: z[k] for k in ("albedo", "material", "normal", "depth")}
    text = run_deferred_shader(open(SHADER).read(), gbuf, z["lights"], z["push"], 64, 36)[..., :3]
    o64 = oracle.deferred(gbuf, z["lights"], z["push"], mode="f64", want=("f64",))
    ok = o64["near_threshold"] == 0
    assert ok.mean() > 0.95
    a = o64["f64"][..., :3]
    assert np.abs(a - text)[ok].max() <= 1e-11 * np.abs(text).max() + 1e-15
    stored = z["out_mix_f64"][..., :3]
    close = np.abs(stored - text) <= 1e-4 * np.abs(text) + 2e-6
    assert close[ok & (z["near_threshold"] == 0)].mean() > 0.995


@pytest.mark.skipif(not os.path.exists(VERT), reason="reference sources not present")
def test_cull_planes_bound_the_clip_volume_of_the_vertex_shaders_position(oracle):
    """The cull's rule is this build's own (the reference's prep_indirect.comp is empty), but the volume it tests
    against is not: a point is inside the six planes of vkr_frustum_planes(vp) exactly when the gl_Position that
    the reference's vertex shader - its text, /root/reference/shaders/offscreen.vert:31 - computes for it lies in
    Vulkan's clip volume (-w <= x, y <= w, 0 <= z <= w).  Checked for model-space points under random model
    matrices, leaving out points within rounding of a plane."""
    from glsl_eval import Interpreter
    synth = importlib.import_module("vk-renderer_amd.synth")
    cam = synth.reference_camera(3840, 2160)
    vp = np.asarray(cam["vp"], np.float64).reshape(4, 4).T  # [row][column]
    planes = np.asarray(cam["planes"], np.float64)
    rng = np.random.default_rng(8)
    mats, points = [], []
    for _ in range(6):
        m = np.eye(4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m[:3, :3] = q * rng.uniform(0.5, 3.0)
        m[:3, 3] = rng.uniform(-40, 40, 3)
        mats.append(m)
    g = {"push": {"vp": vp}, "gl_DrawID": 0, "gl_InstanceIndex": 0, "gl_Position": None,
         "ubo0": {"modelData": [{"matIdx": 0, "modelMatrixBaseIndex": 0}]}, "ubo1": {"modelMatrices": mats},
         "inNormal": np.array([0.0, 0.0, 1.0]), "inTangent": np.array([1.0, 0.0, 0.0]), "inUV": np.zeros(3),
         "outUV": None, "outTexOffset": None, "outTBN": None}
    vert = Interpreter(open(VERT).read(), g, {})
    inside_text, inside_planes, margin = [], [], []
    for k in range(600):
        g["ubo0"]["modelData"][0]["modelMatrixBaseIndex"] = k % len(mats)
        world = np.asarray(cam["pos"], np.float64) + rng.normal(size=3) * np.array([300.0, 200.0, 400.0]) + np.array([0.0, 0.0, -150.0])
        g["inPos"] = np.linalg.solve(mats[k % len(mats)], np.append(world, 1.0))[:3]  # model-space point that lands on `world`
        vert.call("main", [])
        x, y, z, w = g["gl_Position"]
        inside_text.append(-w <= x <= w and -w <= y <= w and 0.0 <= z <= w)
        d = planes[:, :3] @ world + planes[:, 3]
        inside_planes.append(bool((d >= 0).all()))
        margin.append(float(np.abs(d).min() / (np.abs(planes[:, :3]).sum(1).max() * np.abs(world).max() + 1.0)))
    inside_text, inside_planes, margin = np.array(inside_text), np.array(inside_planes), np.array(margin)
    clear = margin > 1e-5
    assert clear.mean() > 0.95 and 0.1 < inside_text[clear].mean() < 0.9
    assert np.array_equal(inside_text[clear], inside_planes[clear])
