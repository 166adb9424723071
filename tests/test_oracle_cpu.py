"""CPU checks of the oracle itself: closed-form known-answer cases for the BRDF pieces of
/root/reference/shaders/deferred.comp, the cull rule, and the committed golden vectors.

The reference has no tests or fixtures, so the oracle is pinned by (a) analytic answers worked out by
hand from the shader's formulas, (b) its own committed outputs, which guard against drift, and (c) in
test_shader_text_cpu.py, the reference's shader text evaluated on the CPU."""
import hashlib
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PI = float(np.float32(3.14159))
INV151 = float(np.frombuffer(np.uint32(0x3F298953).tobytes(), np.float32)[0])


def one_pixel(normal_raw, albedo, rough_u8, metal_u8, depth=0.5):
    g = {"albedo": np.array([[list(albedo) + [255]]], np.uint8),
         "material": np.array([[[77, rough_u8, metal_u8, 255]]], np.uint8),
         "normal": np.array([[list(normal_raw) + [1.0]]], np.float16),
         "depth": np.array([[depth]], np.float32)}
    return g


def identity_push(cam):
    """inverseVP = identity: a 1x1 frame has UV = 0.5 -> clip.xy = 0, so fragPos = (0, 0, depth)."""
    push = np.zeros(20, np.float32)
    push[0:3] = cam
    push[3] = 1.0
    push[4:20] = np.eye(4, dtype=np.float32).reshape(16)
    return push


def light(pos, color):
    rec = np.zeros((1, 12), np.float32)
    rec[0, 0:3] = pos
    rec[0, 4:7] = color
    rec[0, 3] = 0.3   # opening angle, penumbra, direction: must be ignored
    rec[0, 7] = 0.9
    rec[0, 8:11] = (1, 2, 3)
    return rec


def expected_normal_incidence(albedo, rough, metal, light_color, dist):
    """v = l = n = h: vDotN = lDotN = hDotN = vDotH = 1."""
    alb = np.array(albedo, np.float64) / 255.0
    cd = alb * (1 - metal)
    cs = alb * metal
    alpha = max(rough * rough, float(np.float32(0.0064)))
    a2 = alpha * alpha
    ef = (1 - rough) + INV151 * rough
    diffuse = PI * cd * 1.0 * ef  # both Schlick factors are 1 + (f90-1)*0
    D = a2 / ((a2 * a2) * PI)     # d = (a2-1)+1 = a2
    lam = 0.5 * (-1 + np.sqrt(1 + a2 + 0.0))
    G = 1.0 / ((lam + 1) * (lam + 1))
    spec = cs * D * G * 0.25 / 1.0  # F = cs + (1-cs)*0
    li = np.array(light_color, np.float64) / (dist * dist)
    return float(np.float32(0.05)) * alb + li * diffuse + li * spec


@pytest.mark.parametrize("rough_u8,metal_u8", [(128, 0), (255, 255), (0, 255), (10, 0), (64, 128)])
def test_normal_incidence_closed_form(oracle, rough_u8, metal_u8):
    # surface at (0,0,0.5) facing +z (stored normal is negated by the shader), camera and light on the +z axis
    g = one_pixel((0.0, 0.0, -2.0), (200, 100, 50), rough_u8, metal_u8)
    push = identity_push((0.0, 0.0, 4.5))
    L = light((0.0, 0.0, 2.5), (3.0, 2.0, 1.0))
    want = expected_normal_incidence((200, 100, 50), rough_u8 / 255.0, metal_u8 / 255.0, (3.0, 2.0, 1.0), 2.0)
    o64 = oracle.deferred(g, L, push, mode="f64", want=("f64", "rgba8"))
    np.testing.assert_allclose(o64["f64"][0, 0, :3], want, rtol=1e-6)  # albedo/255 etc. in f64 vs fp32 constants
    o32 = oracle.deferred(g, L, push, mode="f32")
    # binary32 in GLSL order loses (a2-1)*hDotN^2+1 near a mirror peak (a2 = 4.1e-5): 6e-4 off at alpha = 0.0064
    np.testing.assert_allclose(o32["f32"][0, 0, :3], want, rtol=2e-3 if rough_u8 <= 20 else 2e-5)
    omx = oracle.deferred(g, L, push, mode="mix", want=("f64",))
    np.testing.assert_allclose(omx["f64"][0, 0, :3], want, rtol=1e-6)
    assert o32["f32"][0, 0, 3] == 1.0
    q = np.rint(np.clip(want, 0, 1) * 255)
    assert np.abs(o32["rgba8"][0, 0, :3].astype(float) - q).max() <= 1 and o32["rgba8"][0, 0, 3] == 255


def test_alpha_floor_applies_below_roughness_0p08(oracle):
    """roughness^2 < 0.0064 => alpha = 0.0064 (deferred.comp:116): roughness 0 and 20/255 give the same specular."""
    push = identity_push((0.0, 0.0, 4.5))
    L = light((0.0, 0.0, 2.5), (1.0, 1.0, 1.0))
    outs = []
    for r in (0, 20):
        g = one_pixel((0.0, 0.0, -1.0), (255, 255, 255), r, 255)
        outs.append(oracle.deferred(g, L, push, mode="f64", want=("f64",))["f64"][0, 0, :3])
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-12)  # metal: diffuse = 0, specular depends on alpha only
    a2 = float(np.float32(0.0064)) ** 2
    lam = 0.5 * (-1 + np.sqrt(1 + a2))
    want = 0.05 + (1 / 4.0) * (a2 / (a2 * a2 * PI)) / ((lam + 1) ** 2) * 0.25
    np.testing.assert_allclose(outs[0], want, rtol=1e-6)


def test_early_out_backfacing_is_ambient_only(oracle):
    push = identity_push((0.0, 0.0, 4.5))
    L = light((0.0, 0.0, 2.5), (50.0, 50.0, 50.0))
    # stored normal +z => shading normal -z: faces away from both camera and light
    g = one_pixel((0.0, 0.0, 1.0), (255, 128, 0), 128, 0)
    o = oracle.deferred(g, L, push, mode="f32")
    np.testing.assert_allclose(o["f32"][0, 0, :3], np.float32(0.05) * np.array([255, 128, 0], np.float32) / np.float32(255), rtol=1e-7)
    # light behind the surface, camera in front: still nothing (lDotN <= 1e-6)
    g = one_pixel((0.0, 0.0, -1.0), (255, 128, 0), 128, 0)
    o = oracle.deferred(g, light((0.0, 0.0, -3.0), (50.0, 50.0, 50.0)), push, mode="f32")
    np.testing.assert_allclose(o["f32"][0, 0, :3], np.float32(0.05) * np.array([255, 128, 0], np.float32) / np.float32(255), rtol=1e-7)


def test_lights_accumulate_and_unused_fields_are_ignored(oracle):
    push = identity_push((0.3, -0.2, 4.5))
    g = one_pixel((0.1, 0.2, -1.0), (90, 180, 30), 100, 40)
    l1, l2 = light((1.0, 0.5, 2.5), (3.0, 2.0, 1.0)), light((-0.7, 0.4, 1.9), (0.5, 4.0, 2.5))
    amb = oracle.deferred(g, np.zeros((0, 12), np.float32), push, mode="f64", want=("f64",))["f64"][0, 0, :3]
    a = oracle.deferred(g, l1, push, mode="f64", want=("f64",))["f64"][0, 0, :3]
    b = oracle.deferred(g, l2, push, mode="f64", want=("f64",))["f64"][0, 0, :3]
    both = oracle.deferred(g, np.concatenate([l1, l2]), push, mode="f64", want=("f64",))["f64"][0, 0, :3]
    np.testing.assert_allclose(both - amb, (a - amb) + (b - amb), rtol=1e-12)
    l1b = l1.copy()
    l1b[0, 3] = 123.0
    l1b[0, 7:12] = -5.0
    a2 = oracle.deferred(g, l1b, push, mode="f64", want=("f64",))["f64"][0, 0, :3]
    assert np.array_equal(a, a2)
    # pointLightRadius is pushed and never read (deferred.comp:8) unless the extension flag is set
    p2 = push.copy()
    p2[3] = 1e-3
    assert np.array_equal(oracle.deferred(g, l1, p2, mode="f64", want=("f64",))["f64"], oracle.deferred(g, l1, push, mode="f64", want=("f64",))["f64"])
    cut = oracle.deferred(g, l1, p2, mode="f64", want=("f64",), radius_cutoff=True)["f64"][0, 0, :3]
    np.testing.assert_allclose(cut, amb, rtol=1e-12)


def test_unprojection_uses_texel_centres_and_column_major_matrix(oracle, synth):
    """A translation in inverseVP's last column must shift fragPos; UV must be (x+0.5)/W."""
    w, h = 4, 2
    g = {"albedo": np.full((h, w, 4), 255, np.uint8), "material": np.full((h, w, 4), 128, np.uint8),
         "normal": np.tile(np.array([0, 0, -1, 1], np.float16), (h, w, 1)), "depth": np.full((h, w), 0.25, np.float32)}
    push = identity_push((0.0, 0.0, 10.0))
    push[4 + 12] = 5.0  # column 3, row 0: x translation
    # a light sitting exactly above pixel (x=1,y=0): clip.x = (1.5/4)*2-1 = -0.25 -> world x = 4.75, y = (0.5/2)*2-1 = -0.5
    L = light((4.75, -0.5, 1.25), (1.0, 1.0, 1.0))
    o = oracle.deferred(g, L, push, mode="f64", want=("f64",))["f64"][..., 0]
    assert np.unravel_index(np.argmax(o), o.shape) == (0, 1)


def test_quantisation_rules(oracle):
    push = identity_push((0.0, 0.0, 4.5))
    # huge light -> saturates to 255; NaN position (zero normal -> normalize(0)) -> 0
    g = one_pixel((0.0, 0.0, -1.0), (255, 255, 255), 128, 0)
    o = oracle.deferred(g, light((0.0, 0.0, 0.6), (1e6, 1e6, 1e6)), push, mode="f32")
    assert tuple(o["rgba8"][0, 0]) == (255, 255, 255, 255)
    g = one_pixel((0.0, 0.0, 0.0), (255, 255, 255), 128, 0)  # the reference's cleared normal (src/main.cpp:128)
    o = oracle.deferred(g, light((0.0, 0.0, 2.0), (1.0, 1.0, 1.0)), push, mode="f32")
    assert np.isnan(o["f32"][0, 0, :3]).all() and tuple(o["rgba8"][0, 0]) == (0, 0, 0, 255)
    # round-half-even on the ambient term only: albedo 51/255*0.05*255 = 2.55 -> 3 ; exact .5 cases
    from oracle import lib
    import ctypes as C
    # 0.5/255 -> 0.5 -> rounds to even 0 ; 1.5/255 -> 2 ; 2.5/255 -> 2
    for val, want in ((0.5 / 255, 0), (1.5 / 255, 2), (2.5 / 255, 2), (-1.0, 0), (2.0, 255)):
        assert int(np.rint(np.clip(np.float64(val), 0, 1) * 255)) == want


def test_cull_rule_known_answers(oracle):
    # planes of the axis-aligned box |x|,|y| <= 1, 0 <= z <= 1 written as clip planes with w = 1:
    planes = np.array([[1, 0, 0, 1], [-1, 0, 0, 1], [0, 1, 0, 1], [0, -1, 0, 1], [0, 0, 1, 0], [0, 0, -1, 1]], np.float32)
    boxes = [
        ((-0.5, -0.5, 0.2), (0.5, 0.5, 0.8), 1),   # inside
        ((-3.0, -0.5, 0.2), (-2.0, 0.5, 0.8), 0),  # left of x >= -1
        ((-3.0, -0.5, 0.2), (-1.0, 0.5, 0.8), 1),  # touches x = -1: d == 0 counts as visible
        ((0.9, 0.9, 0.9), (5.0, 5.0, 5.0), 1),     # straddles three planes
        ((-0.5, -0.5, 1.0001), (0.5, 0.5, 2.0), 0),  # beyond far
        ((-0.5, -0.5, -2.0), (0.5, 0.5, -0.0001), 0),  # before near
        ((-9, -9, -9), (9, 9, 9), 1),              # contains the whole frustum
        ((np.nan, 0, 0), (1, 1, 1), 0),            # NaN corner on the deciding side -> culled
    ]
    n = len(boxes)
    mn = np.zeros((n, 4), np.float32)
    mx = np.zeros((n, 4), np.float32)
    for i, (a, b, _) in enumerate(boxes):
        mn[i, :3], mx[i, :3] = a, b
    mesh = np.array([0, 0, 0, 1, 1, 2, 2, 2], np.uint32)
    mn[:, 3] = mesh.view(np.float32)
    base = np.array([0, 3, 5], np.uint32)
    tmpl = np.zeros(3, oracle.DRAW_CMD_DTYPE)
    tmpl["indexCount"], tmpl["firstIndex"], tmpl["vertexOffset"], tmpl["instanceCount"] = (36, 72, 108), (0, 36, 108), (0, 24, 72), (3, 2, 3)
    res = oracle.cull(mn, mx, planes, base, tmpl)
    assert res["mask"].tolist() == [b[2] for b in boxes]
    assert res["cmds"]["instanceCount"].tolist() == [2, 1, 1]
    assert res["cmds"]["firstInstance"].tolist() == [0, 3, 5]
    assert res["cmds"]["indexCount"].tolist() == [36, 72, 108] and res["cmds"]["vertexOffset"].tolist() == [0, 24, 72]
    assert res["visible"].tolist() == [0, 2, 0xFFFFFFFF, 3, 0xFFFFFFFF, 6, 0xFFFFFFFF, 0xFFFFFFFF]
    assert not res["bad"]


def test_cull_plane_sum_has_no_fused_multiply_add(oracle):
    """A case where fma(n.x, p.x, n.y*p.y) and (n.x*p.x + n.y*p.y) differ in sign of the final d."""
    rng = np.random.default_rng(1)
    found = 0
    for _ in range(20000):
        nx, ny, px, py = rng.standard_normal(4).astype(np.float32) * np.float32(100)
        t1 = np.float32(ny * py)
        plain = np.float32(np.float32(nx * px) + t1)
        fused = np.float32(np.float64(nx) * np.float64(px) + np.float64(t1))
        if plain != fused:
            nw = -plain  # d_plain = 0 (visible), d_fused != 0
            if fused + nw < 0:
                planes = np.zeros((6, 4), np.float32)
                planes[:, 3] = 1.0
                planes[0] = (nx, ny, 0, nw)
                mn = np.array([[px, py, 0, 0]], np.float32)
                mn[0, 3] = np.array([0], np.uint32).view(np.float32)[0]
                mx = mn.copy()
                tmpl = np.zeros(1, oracle.DRAW_CMD_DTYPE)
                res = oracle.cull(mn, mx, planes, np.zeros(1, np.uint32), tmpl)
                assert res["mask"][0] == 1
                found += 1
                if found >= 5:
                    break
    assert found >= 1


def test_aabb_corner_order_and_world_box(oracle):
    pts = np.array([[1, 5, -2], [-3, 2, 4], [0, 0, 0]], np.float32)
    c = oracle.compute_aabb_corners(pts)
    mn, mx = (-3, 0, -2), (1, 5, 4)
    want = [(mn[0], mn[1], mn[2]), (mx[0], mn[1], mn[2]), (mn[0], mx[1], mn[2]), (mn[0], mn[1], mx[2]),
            (mx[0], mx[1], mn[2]), (mx[0], mn[1], mx[2]), (mn[0], mx[1], mx[2]), (mx[0], mx[1], mx[2])]
    assert c.tolist() == [list(map(float, w)) for w in want]  # order of /root/reference/src/vkmodel.cpp:172-173
    model = np.eye(4, dtype=np.float32)
    model[3, :3] = (10, 20, 30)  # column-major translation
    model[0, 0] = 2.0
    lo, hi = oracle.world_aabb(c, model.reshape(16))
    assert lo.tolist() == [4.0, 20.0, 28.0] and hi.tolist() == [12.0, 25.0, 34.0]


def test_golden_manifest_and_oracle_reproduces_it(oracle):
    manifest = json.load(open(os.path.join(GOLDEN, "MANIFEST.json")))
    for name, entries in manifest.items():
        if not name.endswith(".npz"):  # text fixtures: one digest of the file's bytes
            assert hashlib.sha256(open(os.path.join(GOLDEN, name), "rb").read()).hexdigest() == entries["text"], name
            continue
        z = np.load(os.path.join(GOLDEN, name))
        for key, digest in entries.items():
            assert hashlib.sha256(np.ascontiguousarray(z[key]).tobytes()).hexdigest() == digest, (name, key)
    for name in ("deferred_64x36_L13.npz", "deferred_32x18_L32.npz", "deferred_32x18_L256.npz"):
        z = np.load(os.path.join(GOLDEN, name))
        g = {k: z[k] for k in ("albedo", "material", "normal", "depth")}
        o32 = oracle.deferred(g, z["lights"], z["push"], mode="f32")
        omx = oracle.deferred(g, z["lights"], z["push"], mode="mix", want=("f64", "rgba8"))
        assert np.array_equal(o32["f32"], z["out_f32"]) and np.array_equal(o32["rgba8"], z["out_f32_rgba8"])
        assert np.array_equal(omx["f64"], z["out_mix_f64"]) and np.array_equal(omx["rgba8"], z["out_mix_rgba8"])
    z = np.load(os.path.join(GOLDEN, "cull_4096_M16.npz"))
    res = oracle.cull(z["aabb_min"], z["aabb_max"], z["planes"], z["mesh_base"], z["cmd_template"])
    assert res["cmds"].tobytes() == z["cmds"].tobytes() and np.array_equal(res["visible"], z["visible"])


def test_oracle_variants_agree_where_well_conditioned(oracle, synth):
    """f32 (GLSL order), mix (fp32 positions, f64 lighting) and f64 agree to ~1e-4 on most pixels; the few
    that do not are specular peaks of near-mirror surfaces, where binary32 loses the (a2-1)*hDotN^2+1 term."""
    case = synth.deferred_case(96, 54, 64, config_index=11)
    o32 = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32")
    omx = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="mix", want=("f64", "rgba8"))
    a, b = o32["f32"][..., :3].astype(np.float64), omx["f64"][..., :3]
    rel = np.abs(a - b) / (np.abs(b) + 1e-3)
    assert np.median(rel) < 1e-6
    assert (rel > 1e-4).mean() < 0.01
    assert np.abs(o32["rgba8"].astype(int) - omx["rgba8"].astype(int)).max() <= 1


def shader_in_numpy(gbuf, lights, push):
    """A second, independent reading of /root/reference/shaders/deferred.comp:98-131 (and the functions it calls,
    :32-96) in binary64 numpy, vectorised over pixels - written from the GLSL, not from the C restatement."""
    H, W = gbuf["depth"].shape
    # literals of the shader are binary32 constants (glslang folds 1.0/1.51 in binary32 too)
    f = lambda c: float(np.float32(c))
    PI, AMBIENT, ALPHA_MIN, EARLY_OUT, DIST_MIN = f(3.14159), f(0.05), f(0.0064), f(1e-6), f(1e-7)
    INV_151 = float(np.float32(1.0) / np.float32(1.51))
    cam = push[:3].astype(np.float64)
    inv_vp = push[4:20].astype(np.float64).reshape(4, 4).T  # column-major in memory
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    uv = np.stack([(x + 0.5) / W, (y + 0.5) / H], -1)                                           # :99
    clip = np.concatenate([uv * 2 - 1, gbuf["depth"].astype(np.float64)[..., None], np.ones((H, W, 1))], -1)  # :100-101
    world = clip @ inv_vp.T                                                                     # :102
    pos = world[..., :3] / world[..., 3:4]                                                      # :103
    albedo = gbuf["albedo"][..., :3].astype(np.float64) / 255                                   # :105
    color = AMBIENT * albedo                                                                       # :106-107
    nrm = gbuf["normal"][..., :3].astype(np.float64)
    n = -nrm / np.linalg.norm(nrm, axis=-1, keepdims=True)                                      # :108
    material = gbuf["material"][..., :3].astype(np.float64) / 255                               # :109
    rough, metal = material[..., 1:2], material[..., 2:3]                                       # :111-112
    diffuse_color = albedo * (1 - metal)                                                        # :114
    specular_color = albedo * metal                                                             # :115
    alpha = np.maximum(rough * rough, ALPHA_MIN)                                                   # :116
    v = cam - pos
    v /= np.linalg.norm(v, axis=-1, keepdims=True)                                              # :118
    dot = lambda a, b: (a * b).sum(-1, keepdims=True)
    schlick = lambda f0, f90, c: f0 + (f90 - f0) * np.maximum(1 - c, 0) ** 5                    # :32-38
    on_threshold = np.zeros((H, W), bool)
    for rec in lights.astype(np.float64):                                                       # :120
        l = rec[0:3] - pos                                                                      # :121
        dist2 = np.maximum(dot(l, l), DIST_MIN)                                                     # :122
        l = l / np.linalg.norm(l, axis=-1, keepdims=True)                                       # :123
        li = rec[4:7] / dist2                                                                   # :125
        v_n, l_n = dot(v, n), dot(l, n)
        lit = np.minimum(v_n, l_n) > EARLY_OUT                                                       # :43, :84
        on_threshold |= (np.abs(np.minimum(v_n, l_n) - EARLY_OUT) < 1e-9)[..., 0]
        with np.errstate(all="ignore"):
            h = v + l
            h /= np.linalg.norm(h, axis=-1, keepdims=True)                                      # :46, :87
            l_h = dot(v, h)                                                                     # :47 (sic: v)
            f90 = 0.5 * rough + 2.0 * l_h * l_h * rough                                         # :48, :50
            weight = (1 - rough + rough * INV_151) * schlick(1.0, f90, v_n) * schlick(1.0, f90, l_n)  # :49, :52-54
            diff = PI * diffuse_color * l_n * weight                                       # :56
            a2 = alpha * alpha                                                                  # :90
            d = (a2 - 1.0) * dot(h, n) ** 2 + 1.0                                               # :60
            ndf = a2 / (d * d * PI)                                                        # :61
            lam = lambda c: 0.5 * (-1.0 + np.sqrt(1.0 + a2 + np.maximum(1.0 - c * c, 0) / (c * c)))  # :64-70 (c > 0 here)
            g = 1.0 / ((lam(v_n) + 1.0) * (lam(l_n) + 1.0))                                     # :72-76
            spec = schlick(specular_color, 1.0, dot(v, h)) * ndf * g * 0.25 / v_n               # :94-95
        color = color + np.where(lit, li * diff, 0.0)                                           # :127
        color = color + np.where(lit, li * spec, 0.0)                                           # :128
    return color, on_threshold


def test_oracle_agrees_with_an_independent_numpy_reading_of_the_shader(oracle, synth):
    case = synth.deferred_case(48, 27, 20, config_index=9)
    want, on_threshold = shader_in_numpy(case["gbuf"], case["lights"], case["push"])
    got = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f64", want=("f64",))
    skip = on_threshold | got["near_threshold"].astype(bool)
    assert skip.mean() < 0.01
    err = np.abs(got["f64"][..., :3] - want) / np.maximum(np.abs(want), 1e-30)
    err[skip] = 0
    # two binary64 evaluations in different operation orders; D = a2/(d*d*PI) amplifies rounding near mirror peaks
    assert err.max() < 1e-9, err.max()
    assert np.all(got["f64"][..., 3] == 1.0)


def test_cull_oracle_agrees_with_an_independent_numpy_evaluation(oracle, synth):
    """The cull rule in elementwise binary32 numpy (separate multiplies and adds, same association), the compaction
    with boolean indexing: mask, per-mesh counts and the compacted lists must equal the sequential C code."""
    for n, m, idx in ((20000, 37, 0), (65536, 16, 1)):
        scene = synth.cull_case(n, m, config_index=idx) if idx == 0 else synth.make_adversarial_cull(n, synth.SEED_CULL + 99, synth.reference_camera(3840, 2160), m)
        planes = (scene["planes"] if idx == 0 else synth.reference_camera(3840, 2160)["planes"]).astype(np.float32).reshape(6, 4)
        mn, mx = scene["aabb_min"][:, :3], scene["aabb_max"][:, :3]
        vis = np.ones(n, bool)
        for nx, ny, nz, nw in planes:
            px = mx[:, 0] if nx >= 0 else mn[:, 0]
            py = mx[:, 1] if ny >= 0 else mn[:, 1]
            pz = mx[:, 2] if nz >= 0 else mn[:, 2]
            d = ((np.float32(nx) * px + np.float32(ny) * py) + np.float32(nz) * pz) + np.float32(nw)
            assert d.dtype == np.float32
            vis &= d >= 0
        res = oracle.cull(scene["aabb_min"], scene["aabb_max"], planes, scene["mesh_base"], scene["cmd_template"])
        assert np.array_equal(res["mask"].astype(bool), vis)
        mesh = scene["aabb_min"][:, 3].view(np.uint32)
        assert np.array_equal(res["cmds"]["instanceCount"], np.bincount(mesh[vis], minlength=m).astype(np.uint32))
        assert np.array_equal(res["cmds"]["firstInstance"], scene["mesh_base"][:m])
        for k in range(m):
            b, c = int(scene["mesh_base"][k]), int(res["cmds"]["instanceCount"][k])
            assert np.array_equal(res["visible"][b:b + c], np.flatnonzero(vis & (mesh == k)).astype(np.uint32))


def test_oracle_constants_are_the_shaders():
    """The numbers the CPU restatement uses, against the text of the reference's shader where that is present (this
    container; the GPU box has no /root/reference): every floating literal of deferred.comp appears in the oracle's
    sources as the same binary32 value, 1.0/1.51 as the constant it folds to, and the light table's length and
    record layout are the ones the oracle's callers assume."""
    import re
    import struct
    shader = "/root/reference/shaders/deferred.comp"
    if not os.path.exists(shader):
        pytest.skip("reference sources not present")
    strip = lambda text: re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", text, flags=re.S))
    literal = r"(?<![\w.])(\d+\.\d*(?:[eE][-+]?\d+)?|\d+[eE][-+]?\d+)f?"
    glsl = strip(open(shader).read())
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    ours = strip(open(os.path.join(root, "deferred_oracle.c")).read() + open(os.path.join(root, "deferred_oracle_body.inc")).read())
    f32 = lambda text: struct.unpack("<f", struct.pack("<f", float(text)))[0]
    theirs = {f32(x) for x in re.findall(literal, glsl)}
    mine = {f32(x) for x in re.findall(literal, ours)}
    assert {f32("0.0064"), f32("0.05"), f32("1e-6"), f32("1e-7"), f32("3.14159"), f32("0.25"), f32("1.51")} <= theirs  # the shader as surveyed
    missing = theirs - mine - {f32("1.51"), 0.0, 2.0}  # 1.51 only ever divides 1.0; 0.0 and 2.0 are spelled as integers or folded
    assert not missing, missing
    # 1.0 / 1.51 in binary32, as a GLSL front end folds it, is the bit pattern the oracle (and the kernel) carry
    folded = struct.unpack("<I", struct.pack("<f", np.float32(1.0) / np.float32(1.51)))[0]
    assert folded == 0x3F298953 and "0x3F298953" in ours
    # the light table: 13 records, and a record of four vec3 / float pairs = 48 bytes under std140
    assert re.search(r"#define\s+LIGHT_COUNT\s+13\b", glsl) and re.search(r"\blights\s*\[\s*LIGHT_COUNT\s*\]", glsl)
    fields = re.search(r"struct\s+Light\s*\{([^}]*)\}", glsl).group(1)
    names = re.findall(r"\b(vec3|float)\s+(\w+)\s*;", fields)
    assert [t for t, _ in names] == ["vec3", "float"] * 3 and [n for _, n in names][:5] == ["pos", "opening_angle", "color", "penumbra_angle", "dir"]
    # -> pos at byte 0, colour at 16, 48 bytes per record: what vk-renderer_amd.synth writes and the oracle reads
    synth = __import__("importlib").import_module("vk-renderer_amd.synth")
    lights = synth.deferred_case(8, 8, 13, config_index=0)["lights"]
    assert lights.dtype == np.float32 and lights.reshape(13, -1).shape[1] * 4 == 48
