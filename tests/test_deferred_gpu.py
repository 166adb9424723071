"""GPU parity of the deferred resolve against the CPU oracle, through the C ABI.

Bars (north_star): float colour within 1e-4 relative, 8-bit colour within 1 LSB.
  * `mix` oracle  = lighting evaluated in binary64 on the reference's binary32 positions: the GPU
                    must be within 1e-4 everywhere (pixels sitting on the 1e-6 early-out threshold
                    are excluded: the specular term is discontinuous there), plus, per channel, twice
                    what one binary32 ulp of the view direction moves that channel by (the oracle's
                    conditioning probe; next to nothing except at GGX peaks of near-mirror surfaces,
                    1-2 channels per million, whose number beyond the plain 1e-4 is bounded as well).
  * `f32` oracle  = binary32 in GLSL evaluation order: compared, in float and in 8 bits, where that
                    evaluation is itself well-conditioned (within 2e-5 of `mix`).
"""
import numpy as np
import pytest

from _util import ABS_TOL, REL_TOL, assert_color_close, rel_err, run_deferred_gpu

pytestmark = pytest.mark.gpu


def _oracles(oracle, case, radius=False, rows=None):
    o32 = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", radius_cutoff=radius, rows=rows, threads=8)
    omx = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="mix", radius_cutoff=radius, rows=rows, threads=8,
                          want=("f64", "rgba8", "view_sensitivity"))
    return o32, omx


# Share of the colour channels that may lie beyond the plain 1e-4 of the `mix` oracle (and then within the bound
# widened by the conditioning probe).  Measured: 0 - 3 channels of the 6.2 million of the 1080p x 32 frame, 2 - 5 of
# the 3.1 million of the first 270 rows of the 4K x 256 frame, over four builds that differ only in rounding
# (tools/parity_stats.py); which channels they are changes with every rounding-level change of the kernel.
def _allowed_beyond_plain_bound(pixels: int) -> int:
    """One pixel (three channels) per started million pixels of a large frame; none in the small cases."""
    return 3 * -(-pixels // 1_000_000) if pixels >= 100_000 else 0


def _well_conditioned(o32, omx, sl, lights, what):
    """Pixels where the binary32 GLSL-order evaluation is itself within 2e-5 of the binary64 one, i.e. where it can
    serve as a reference at 1e-4.  The share of such pixels falls with the number of lights (each is a chance of a
    light a fraction of a unit from a far surface, or of a mirror peak at the alpha floor).  Measured over the cases
    of this file: 0.9996-1 at up to 13 lights, 0.998-0.999 at 32-64, 0.984-0.994 at 256, 0.975 at 1500; the floors
    below sit within 1 % of those, so a change that makes the GPU comparison skip more pixels fails here."""
    well = rel_err(o32["f32"][sl][..., :3], omx["f64"][sl][..., :3]).max(-1) <= 2e-5
    floor = 0.985 if lights <= 64 else (0.974 if lights <= 256 else 0.965)
    assert well.mean() >= floor, f"{what}: only {well.mean():.4f} of the pixels have a usable binary32 oracle value (floor {floor})"
    return well


def _check_float(got, o32, omx, what, lights, rows=None):
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    g = got[sl][..., :3]
    near = omx["near_threshold"][sl]
    assert near.mean() < 1e-3
    sens = omx["view_sensitivity"][sl]
    assert_color_close(g, omx["f64"][sl][..., :3], near, what + " vs mix oracle", view_sensitivity=sens)
    ref = omx["f64"][sl][..., :3]
    plain = (np.abs(g.astype(np.float64) - ref) > REL_TOL * np.abs(ref) + ABS_TOL) & ~near.astype(bool)[..., None]
    assert plain.sum() <= _allowed_beyond_plain_bound(plain.size // 3), f"{what}: {int(plain.sum())} of {plain.size} channels beyond the plain 1e-4"
    if plain.any():
        print(f"{what}: {int(plain.sum())} of {plain.size} channels beyond the plain 1e-4 of the mix oracle (largest {rel_err(g, ref)[plain].max():.2e}), all within the widened bound")
    well = _well_conditioned(o32, omx, sl, lights, what)
    assert_color_close(g, o32["f32"][sl][..., :3], near | ~well, what + " vs f32 oracle", view_sensitivity=sens)
    assert np.all(got[sl][..., 3] == 1.0)


def _check_u8(got, o32, omx, what, lights, rows=None):
    sl = slice(None) if rows is None else slice(rows[0], rows[1])
    near = omx["near_threshold"][sl].astype(bool)
    # the binary32 GLSL-order evaluation is compared where it is itself well-conditioned (see _check_float): a light a
    # fraction of a unit from a far surface moves its own 8-bit result by more than one step
    well = _well_conditioned(o32, omx, sl, lights, what)
    for name, ref, skip in (("mix", omx["rgba8"], near), ("f32", o32["rgba8"], near | ~well)):
        d = np.abs(got[sl].astype(np.int32) - ref[sl].astype(np.int32))
        d[skip] = 0
        assert d.max() <= 1, f"{what}: {int((d > 1).sum())} channels differ by more than 1 LSB from the {name} oracle (max {d.max()})"
    assert np.all(got[sl][..., 3] == 255)


def _distance_from_the_literal_binary32_run(got, got8, o32, near, what, max_float_share, max_lsb_share):
    """Every pixel counted, none masked for conditioning (only those on the 1e-6 early-out, where a contribution may
    legitimately be there or not): the share of colour channels whose float value is more than 1e-4 from the
    binary32 evaluation in GLSL order, and the share whose 8-bit value is more than one step from that evaluation's
    8-bit value.  The kernels are built to follow the binary64 result (which the `mix` comparison holds them to
    everywhere); this is how far that takes them from a literal binary32 run, with an upper bound."""
    keep = ~near.astype(bool)
    f = rel_err(got[..., :3], o32["f32"][..., :3])[keep]
    float_share = float((f > REL_TOL).mean())
    out = {"float_share": float_share, "float_max": float(f.max())}
    if got8 is not None:
        d = np.abs(got8[..., :3].astype(np.int32) - o32["rgba8"][..., :3].astype(np.int32))[keep]
        out.update(lsb_share=float((d > 1).mean()), lsb_max=int(d.max()))
        assert out["lsb_share"] <= max_lsb_share, (what, out)
    assert float_share <= max_float_share, (what, out)
    print(f"{what}: distance from the literal binary32 run: {out}")
    return out


@pytest.mark.parametrize("w,h,lights", [(64, 36, 13), (32, 18, 32), (32, 18, 256), (37, 19, 5), (4, 1, 1), (1, 1, 3),
                                        (130, 33, 2), (1000, 8, 13)])
def test_float_parity(vkr, synth, oracle, ctx, w, h, lights):
    case = synth.deferred_case(w, h, lights, config_index=7)
    o32, omx = _oracles(oracle, case)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    _check_float(got, o32, omx, f"{w}x{h} L={lights}", lights)


@pytest.mark.parametrize("w,h,lights", [(64, 36, 13), (37, 19, 32), (128, 72, 256)])
def test_rgba8_parity(vkr, synth, oracle, ctx, w, h, lights):
    case = synth.deferred_case(w, h, lights, config_index=3)
    o32, omx = _oracles(oracle, case)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM)
    _check_u8(got, o32, omx, f"{w}x{h} L={lights}", lights)


def test_reference_sized_frame_1080p_32_lights(vkr, synth, oracle, ctx):
    """The reference's own runnable size (1920x1080, 32 point lights): the whole frame against both oracle
    evaluations, in float and in the 8-bit format the reference writes."""
    case = synth.deferred_case(1920, 1080, 32, config_index=0)
    o32, omx = _oracles(oracle, case)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    _check_float(got, o32, omx, "1920x1080 L=32", 32)
    got8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM)
    _check_u8(got8, o32, omx, "1920x1080 L=32", 32)
    _distance_from_the_literal_binary32_run(got, got8, o32, omx["near_threshold"], "1920x1080 L=32", 5e-4, 1e-6)  # measured: 2.5e-4 (largest 1.1 %), 1.6e-7 (2 steps)


def test_rgba16f_output(vkr, synth, oracle, ctx):
    case = synth.deferred_case(64, 36, 13, config_index=5)
    _, omx = _oracles(oracle, case)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R16G16B16A16_SFLOAT)
    ref = np.clip(omx["f64"][..., :3], -65504, 65504).astype(np.float16)
    # one half-precision ulp (the float result may sit on either side of a rounding boundary)
    assert np.all(np.abs(got[..., :3].astype(np.float64) - ref.astype(np.float64)) <= np.spacing(np.abs(ref)).astype(np.float64))
    assert np.all(got[..., 3] == np.float16(1.0))


def test_bgra8_output_is_the_rgba8_output_with_red_and_blue_swapped(vkr, synth, ctx):
    """B8G8R8A8_UNORM is the format of the images the reference binds as outTex (its swapchain images,
    /root/reference/src/vkpresent.cpp:17): byte 0 of a texel holds blue.  Whole-table and tile-list paths."""
    case = synth.deferred_case(130, 70, 40, config_index=2)
    case["push"] = case["push"].copy()
    case["push"][3] = 60.0
    for flags in (0, vkr.DEFERRED_RADIUS_CUTOFF):
        rgba, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM, flags=flags)
        bgra, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_B8G8R8A8_UNORM, flags=flags)
        assert len(np.unique(rgba[..., 0].astype(int) - rgba[..., 2])) > 1  # red and blue do differ in this frame
        assert np.array_equal(bgra[..., [2, 1, 0, 3]], rgba)


def test_device_self_test(ctx):
    """The resolve's shared-reciprocal division must equal IEEE division bit for bit (positions are
    reproduced exactly; a single ulp there is visible at 1e-4 for lights close to far surfaces)."""
    assert ctx.self_test(1 << 24) == 0


def test_zero_lights_is_ambient(vkr, synth, ctx):
    case = synth.deferred_case(40, 10, 0)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    amb = np.float32(0.05) * (case["gbuf"]["albedo"][..., :3].astype(np.float32) / np.float32(255))
    np.testing.assert_allclose(got[..., :3], amb, rtol=3e-7, atol=0)


def test_kernels_against_the_shader_text_fixture(vkr, ctx):
    """tests/golden/shader_text_eval_24x10_L13.npz holds a frame and the colours that the reference's shader - its
    own text, evaluated in binary64 by tests/glsl_eval.py (tests/golden/make_shader_eval.py) - gives for it.  The
    float outputs within 1e-4 of them, the 8-bit outputs within one step of their quantisation, both workgroup
    shapes."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shader_text_eval_24x10_L13.npz"))
    case = {"width": 24, "height": 10, "gbuf": {k: z[k] for k in ("albedo", "material", "normal", "depth")}, "lights": z["lights"], "push": z["push"]}
    want = z["color"][..., :3]
    for flags in (0, vkr.DEFERRED_WHOLE_TILE_WORKGROUPS):
        got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags)
        assert_color_close(got[..., :3], want, what="rgba32f against the shader text")
        assert (got[..., 3] == 1.0).all()
        half, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R16G16B16A16_SFLOAT, flags=flags)
        assert np.abs(half[..., :3].astype(np.float64) - want).max() <= 1e-3 * want.max()
        q = np.rint(np.clip(want, 0, 1) * 255)
        for fmt, order in ((vkr.FORMAT_R8G8B8A8_UNORM, (0, 1, 2)), (vkr.FORMAT_B8G8R8A8_UNORM, (2, 1, 0))):
            got8, _ = run_deferred_gpu(vkr, ctx, case, fmt, flags=flags)
            assert np.abs(got8[..., list(order)].astype(np.float64) - q).max() <= 1 and (got8[..., 3] == 255).all()


@pytest.mark.parametrize("name", ["shader_text_mirror_peak_64x8_L13.npz", "shader_text_near_light_32x8_L13.npz"])
def test_kernels_on_the_hard_frames_of_the_shader_text(vkr, ctx, name):
    """tests/golden/make_shader_eval_hard.py: the reference's text, in binary64, on (a) a smooth metallic wall with
    lights on the mirror direction - the GGX peak at the floor of alpha, where the text's own binary32 evaluation is
    off by up to 0.8 % - and (b) lights 0.05-0.5 units from surfaces 320-384 units out.  Positions reconstruct
    exactly in both frames, so the binary64 colours are the target: float output within 1e-4 on every channel of
    every pixel, for every workgroup shape and through the light lists."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    h, w = z["depth"].shape
    case = {"width": w, "height": h, "gbuf": {k: z[k] for k in ("albedo", "material", "normal", "depth")}, "lights": z["lights"], "push": z["push"]}
    want = z["color"][..., :3]
    for flags in (0, vkr.DEFERRED_WHOLE_TILE_WORKGROUPS, vkr.DEFERRED_RADIUS_CUTOFF, vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS):
        got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags)  # (the radius is 1e6: no light is cut)
        assert_color_close(got[..., :3], want, what=f"{name} flags {flags}")
        worst = (np.abs(got[..., :3].astype(np.float64) - want) / (np.abs(want) + 2e-2)).max()
        print(f"{name} flags {flags}: max relative distance from the binary64 text {worst:.2e}")
    got8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM)
    assert np.abs(got8[..., :3].astype(np.float64) - np.rint(np.clip(want, 0, 1) * 255)).max() <= 1


def test_band_rows_match_full_frame(vkr, synth, ctx):
    """Screen-band sharding: shading rows [a,b) alone gives the same bytes as the full dispatch."""
    case = synth.deferred_case(96, 70, 32, config_index=2)
    full, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    for rows in ((0, 9), (9, 41), (41, 70)):
        band, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, rows=rows)
        assert np.array_equal(band[rows[0]:rows[1]], full[rows[0]:rows[1]])
        rest = np.ones(70, bool)
        rest[rows[0]:rows[1]] = False
        assert np.all(band[rest] == 0)  # rows outside the band are untouched


@pytest.mark.parametrize("w,h,lights", [(64, 36, 13), (131, 77, 200), (37, 9, 700), (5, 3, 1), (256, 250, 129)])
def test_workgroup_shapes_give_the_same_bytes(vkr, synth, ctx, w, h, lights):
    """A launch of a few rounds of workgroups (small frames, screen bands) runs as single-wave workgroups over
    32 x 8 pixel tiles, larger ones as four-wave workgroups over 32 x 32; VKR_DEFERRED_WHOLE_TILE_WORKGROUPS forces
    the latter.  Same pixels, same light order: every output format, and the whole-table loop with the radius rule,
    must come out byte for byte the same, on extents that are no multiple of either tile."""
    case = synth.deferred_case(w, h, lights, config_index=5)
    for fmt in (vkr.FORMAT_R32G32B32A32_SFLOAT, vkr.FORMAT_R16G16B16A16_SFLOAT, vkr.FORMAT_R8G8B8A8_UNORM, vkr.FORMAT_B8G8R8A8_UNORM):
        fine, _ = run_deferred_gpu(vkr, ctx, case, fmt)
        coarse, _ = run_deferred_gpu(vkr, ctx, case, fmt, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
        assert np.array_equal(fine, coarse), fmt
    case["push"] = case["push"].copy()
    case["push"][3] = 80.0
    flags = vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS
    fine, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags)
    coarse, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags | vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
    assert np.array_equal(fine, coarse)
    r0, r1 = h // 3, h - h // 4
    band, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags, rows=(r0, r1))
    assert np.array_equal(band[r0:r1], coarse[r0:r1])


@pytest.mark.parametrize("w,h,lights,radius", [(64, 36, 64, 150.0), (96, 40, 700, 60.0), (70, 33, 1500, 1e6), (33, 70, 300, 5.0)])
def test_radius_cutoff_extension(vkr, synth, oracle, ctx, w, h, lights, radius):
    """Opt-in extension (not in the reference): a light contributes only where dist^2 < radius^2.  The kernel
    builds per-tile light lists in LDS; the result must equal the per-pixel rule of the oracle, and must be
    bit-identical to the same kernel looping over the whole table (lists are conservative and ordered)."""
    case = synth.deferred_case(w, h, lights, config_index=4)
    case["push"] = case["push"].copy()
    case["push"][3] = radius
    o32, omx = _oracles(oracle, case, radius=True)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=vkr.DEFERRED_RADIUS_CUTOFF)
    _check_float(got, o32, omx, "radius cutoff", lights)
    flat, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT,
                               flags=vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS)
    assert np.array_equal(got, flat)
    if radius < 1e5:
        # the cutoff really removes light: the unrestricted result is brighter somewhere
        ref_all, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
        assert (ref_all[..., :3] > got[..., :3] * 1.01).any()
    else:
        ref_all, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
        assert np.array_equal(ref_all, got)  # a radius beyond every light changes nothing


@pytest.mark.parametrize("w,h,lights,radius", [(300, 200, 600, 40.0), (257, 129, 300, 60.0), (640, 360, 5000, 25.0)])
def test_radius_cutoff_with_coarse_bins(vkr, synth, ctx, w, h, lights, radius):
    """From 256 lights up the tiled resolve first lists, per 128 x 128 pixel bin, the lights that can reach the
    bin's view pyramid, and tiles test only those.  Several bins, partial bins at the right and bottom edges,
    more lights than one pass tests (4096), and screen bands: always the bytes of the whole-table loop."""
    case = synth.deferred_case(w, h, lights, config_index=6)
    case["push"] = case["push"].copy()
    case["push"][3] = radius
    flat, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT,
                               flags=vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=vkr.DEFERRED_RADIUS_CUTOFF)
    assert np.array_equal(got, flat)
    lit = (got[..., :3] > np.float32(0.05) * (case["gbuf"]["albedo"][..., :3].astype(np.float32) / np.float32(255)) * 1.001).any(-1)
    assert 0.001 < lit.mean() < 0.999  # the cutoff leaves some pixels with and some without any light
    r0, r1 = h // 3 + 1, h - 7
    band, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=vkr.DEFERRED_RADIUS_CUTOFF, rows=(r0, r1))
    assert np.array_equal(band[r0:r1], flat[r0:r1])
    # the other output formats: the list path's bytes are those of the whole-table loop, and the half output is the
    # binary32 colour converted (not a colour whose last multiply-add was rounded straight to half)
    for fmt in (vkr.FORMAT_R16G16B16A16_SFLOAT, vkr.FORMAT_R8G8B8A8_UNORM, vkr.FORMAT_B8G8R8A8_UNORM):
        a, _ = run_deferred_gpu(vkr, ctx, case, fmt, flags=vkr.DEFERRED_RADIUS_CUTOFF)
        b, _ = run_deferred_gpu(vkr, ctx, case, fmt, flags=vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS)
        assert np.array_equal(a, b), fmt
        if fmt == vkr.FORMAT_R16G16B16A16_SFLOAT:
            with np.errstate(over="ignore"):
                assert np.array_equal(a.view(np.uint16), got.astype(np.float16).view(np.uint16))


def test_depth_grouped_tiles(vkr, synth):
    """A tile whose pixels span more depth than the radius has them sorted into eight depth groups, each with a
    list of its own.  Frames of such tiles (depth uncorrelated between neighbours) in every output format, a band
    of rows, a light cluster that overflows one group's list (that group falls back on the tile's list), and the
    decision the library takes from the frame before (a frame without deep tiles, then frames with them, then
    without again): always the bytes of the whole-table loop."""
    ctx = vkr.Context(0)
    tiled, flat_flags = vkr.DEFERRED_RADIUS_CUTOFF, vkr.DEFERRED_RADIUS_CUTOFF | vkr.DEFERRED_NO_TILE_LISTS
    w, h = 416, 200  # 13 x 7 tiles, the last row and column partial
    noisy = synth.deferred_case(w, h, 3000, config_index=9)
    noisy["push"] = noisy["push"].copy()
    noisy["push"][3] = 25.0
    smooth = dict(noisy, gbuf=dict(noisy["gbuf"], depth=np.full((h, w), 0.995, np.float32)))
    flat = {}
    for name, case in (("noisy", noisy), ("smooth", smooth)):
        flat[name], _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flat_flags)
    deep_seen = []
    for name in ("smooth", "noisy", "noisy", "noisy", "smooth", "smooth", "noisy"):
        got, _ = run_deferred_gpu(vkr, ctx, {"noisy": noisy, "smooth": smooth}[name], vkr.FORMAT_R32G32B32A32_SFLOAT, flags=tiled)
        assert np.array_equal(got, flat[name]), name
        deep_seen.append(ctx.tile_stats()["deep_tiles"])
    assert deep_seen[0] == 0 and deep_seen[4] == 0 and deep_seen[5] == 0
    assert deep_seen[2] > 20 and deep_seen[3] == deep_seen[2]  # by its second frame at the latest the noisy scene is shaded in depth groups
    for fmt in (vkr.FORMAT_R16G16B16A16_SFLOAT, vkr.FORMAT_R8G8B8A8_UNORM, vkr.FORMAT_B8G8R8A8_UNORM):
        a, _ = run_deferred_gpu(vkr, ctx, noisy, fmt, flags=tiled)
        a, _ = run_deferred_gpu(vkr, ctx, noisy, fmt, flags=tiled)
        assert ctx.tile_stats()["deep_tiles"] > 20
        b, _ = run_deferred_gpu(vkr, ctx, noisy, fmt, flags=flat_flags)
        assert np.array_equal(a, b), fmt
    band, _ = run_deferred_gpu(vkr, ctx, noisy, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=tiled, rows=(37, 150))
    assert np.array_equal(band[37:150], flat["noisy"][37:150]) and not band[:37].any() and not band[150:].any()
    # 200 lights within a unit of one pixel's position: its depth group lists more than a group's list holds
    crowd = dict(noisy, lights=noisy["lights"][:500].copy())
    pos = synth.unproject(noisy["camera"]["inv_vp"], w, h, noisy["gbuf"]["depth"])[100, 210]
    rng = np.random.default_rng(5)
    crowd["lights"][:200, 0:3] = (pos + rng.uniform(-1, 1, (200, 3))).astype(np.float32)
    want, _ = run_deferred_gpu(vkr, ctx, crowd, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flat_flags)
    for _ in range(2):
        got, _ = run_deferred_gpu(vkr, ctx, crowd, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=tiled)
        assert np.array_equal(got, want)
    st = ctx.tile_stats()
    assert st["deep_tiles"] > 0 and st["max_list"] >= 200 and st["overflow_tiles"] == 0
    # texels that are not numbers inside deep tiles (the depth order and the groups' depth ranges are built from the
    # very values that are broken here): depth NaN / +inf / negative, cleared normals (0, 0, 0), a NaN normal - the
    # same bits as the whole-table loop, NaN payloads included, and the same 8-bit frame
    broken = dict(noisy, gbuf={k: v.copy() for k, v in noisy["gbuf"].items()})
    rng = np.random.default_rng(11)
    ys, xs = rng.integers(0, h, 400), rng.integers(0, w, 400)
    broken["gbuf"]["depth"][ys[:100], xs[:100]] = np.nan
    broken["gbuf"]["depth"][ys[100:150], xs[100:150]] = np.inf
    broken["gbuf"]["depth"][ys[150:200], xs[150:200]] = -0.5
    broken["gbuf"]["normal"][ys[200:300], xs[200:300], :3] = 0
    broken["gbuf"]["normal"][ys[300:400], xs[300:400], 1] = np.nan
    broken["gbuf"]["depth"][64:96, 32:64] = np.nan  # a whole tile without a single depth that is a number
    for fmt, as_bits in ((vkr.FORMAT_R32G32B32A32_SFLOAT, np.uint32), (vkr.FORMAT_R8G8B8A8_UNORM, np.uint8)):
        want, _ = run_deferred_gpu(vkr, ctx, broken, fmt, flags=flat_flags)
        for _ in range(2):
            got, _ = run_deferred_gpu(vkr, ctx, broken, fmt, flags=tiled)
            assert np.array_equal(got.view(as_bits), want.view(as_bits)), fmt
        if as_bits is np.uint32:
            assert np.isnan(want[..., :3]).any() and ctx.tile_stats()["deep_tiles"] > 20


def test_radius_argument_range(vkr, synth, ctx):
    case = synth.deferred_case(16, 4, 1)
    gb = vkr.GBuffer(ctx, 16, 4).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 16, 4)
    for bad in (0.0, -1.0, 1e9, float("nan")):
        push = case["push"].copy()
        push[3] = bad
        with pytest.raises(vkr.VkrError, match="pointLightRadius"):
            ctx.deferred_resolve(vkr.make_deferred_args(gb, out, lights, 1, push, flags=vkr.DEFERRED_RADIUS_CUTOFF))
    push = case["push"].copy()
    push[3] = float("inf")
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out, lights, 1, push, flags=vkr.DEFERRED_RADIUS_CUTOFF))
    ctx.wait_idle()


def test_golden_fixture(vkr, ctx):
    import os
    path = os.path.join(os.path.dirname(__file__), "golden", "deferred_64x36_L13.npz")
    z = np.load(path)
    case = {"width": 64, "height": 36, "gbuf": {k: z[k] for k in ("albedo", "material", "normal", "depth")},
            "lights": z["lights"], "push": z["push"]}
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    assert_color_close(got[..., :3], z["out_mix_f64"][..., :3], z["near_threshold"], "golden 64x36 L13")
    got8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM)
    assert np.abs(got8.astype(int) - z["out_f32_rgba8"].astype(int)).max() <= 1


def test_full_4k_properties(vkr, synth, oracle, ctx):
    """BASELINE size (3840x2160, 256 lights): sampled rows against the oracle plus size-independent
    properties: linearity in the light colours and invariance to the light order."""
    case = synth.deferred_case(3840, 2160, 256, config_index=1)
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
    assert np.isfinite(got).all()
    got8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM)
    # the per-pixel light lists (the default at this light count) against the loop that evaluates every light on every pixel
    loop, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
    assert np.array_equal(got.view(np.uint8), loop.view(np.uint8))
    loop8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
    assert np.array_equal(got8, loop8)
    del loop, loop8
    # and a table of 1024 lights, which the lists take in four chunks
    c4 = dict(case, lights=synth.make_lights(1024, 77, case["camera"], case["gbuf"]))
    lists4, _ = run_deferred_gpu(vkr, ctx, c4, vkr.FORMAT_R16G16B16A16_SFLOAT)
    loop4, _ = run_deferred_gpu(vkr, ctx, c4, vkr.FORMAT_R16G16B16A16_SFLOAT, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
    assert np.array_equal(lists4.view(np.uint8), loop4.view(np.uint8))
    del lists4, loop4
    for rows in ((0, 2), (1079, 1081), (2158, 2160)):
        o32, omx = _oracles(oracle, case, rows=rows)
        _check_float(got, o32, omx, f"4K rows {rows}", 256, rows=rows)
        sl = slice(rows[0], rows[1])
        _distance_from_the_literal_binary32_run(got[sl], got8[sl], {k: v[sl] for k, v in o32.items() if hasattr(v, "shape") and v.shape[:1] == got.shape[:1]},
                                                omx["near_threshold"][sl], f"4K L=256 rows {rows}", 4e-3, 1e-4)  # measured: 1.0e-3 - 2.0e-3 (largest 0.6 %), none
    amb = 0.05 * case["gbuf"]["albedo"][..., :3].astype(np.float64) / 255.0
    # linearity: doubling every light colour doubles (colour - ambient)
    c2 = dict(case)
    c2["lights"] = case["lights"].copy()
    c2["lights"][:, 4:7] *= 2.0
    got2, _ = run_deferred_gpu(vkr, ctx, c2, vkr.FORMAT_R32G32B32A32_SFLOAT)
    lit1 = got[..., :3].astype(np.float64) - amb
    lit2 = got2[..., :3].astype(np.float64) - amb
    assert np.all(np.abs(lit2 - 2 * lit1) <= 2e-6 * np.abs(lit2) + 1e-6)
    # permutation of the light table only changes the summation order
    c3 = dict(case)
    c3["lights"] = case["lights"][::-1].copy()
    got3, _ = run_deferred_gpu(vkr, ctx, c3, vkr.FORMAT_R32G32B32A32_SFLOAT)
    assert np.all(np.abs(got3[..., :3].astype(np.float64) - got[..., :3]) <= 2e-5 * np.abs(got[..., :3]) + 1e-6)
    # 8-bit frame: quantisation of the float frame, within 1 LSB
    q = np.rint(np.clip(got[..., :3].astype(np.float64), 0, 1) * 255)
    assert np.abs(got8[..., :3].astype(np.float64) - q).max() <= 1


def test_argument_errors(vkr, synth, ctx):
    case = synth.deferred_case(16, 4, 1)
    gb = vkr.GBuffer(ctx, 16, 4).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 16, 4)
    a = vkr.make_deferred_args(gb, out, lights, 1, case["push"])
    a.albedo.width = 15
    with pytest.raises(vkr.VkrError, match="extent"):
        ctx.deferred_resolve(a)
    a = vkr.make_deferred_args(gb, out, lights, 1, case["push"])
    a.normal.pitch_bytes = 64
    with pytest.raises(vkr.VkrError, match="pitch"):
        ctx.deferred_resolve(a)
    a = vkr.make_deferred_args(gb, out, lights, 1, case["push"])
    a.depth.format = vkr.FORMAT_R8G8B8A8_UNORM
    with pytest.raises(vkr.VkrError, match="format"):
        ctx.deferred_resolve(a)
    a = vkr.make_deferred_args(gb, out, lights, 1, case["push"])
    a.lights = None
    with pytest.raises(vkr.VkrError, match="lights"):
        ctx.deferred_resolve(a)
    a = vkr.make_deferred_args(gb, out, lights, 1, case["push"])
    a.flags = 0x80
    with pytest.raises(vkr.VkrError, match="flags"):
        ctx.deferred_resolve(a)


@pytest.mark.parametrize("radius", [None, 1000.0, 30.0])
def test_not_a_number_pixels(vkr, synth, oracle, ctx, radius):
    """Cleared texels (normal 0 -> normalize(0)), NaN / Inf normals and depths: NaN wherever the reference
    arithmetic yields NaN (stored as 0 in rgba8), the ambient colour where no light takes part."""
    w, h, L = 64, 24, 9
    case = synth.deferred_case(w, h, L, config_index=11)
    g = {k: v.copy() for k, v in case["gbuf"].items()}
    g["normal"][2, 5:9, :3] = 0.0                      # cleared attachment (main.cpp:128)
    g["normal"][3, 7, 0] = np.float16(np.nan)
    g["normal"][4, 8, 1] = np.float16(np.inf)
    g["depth"][5, 9] = np.nan
    g["depth"][6, 10] = np.inf
    g["normal"][7, 40:44, :3] = 0.0
    g["albedo"][7, 40:44, :3] = 200                    # nonzero albedo: ambient alone would be visible
    case = dict(case, gbuf=g)
    flags = 0
    if radius:
        push = case["push"].copy()
        push[3] = radius
        case["push"] = push
        flags = vkr.DEFERRED_RADIUS_CUTOFF
    ref = oracle.deferred(g, case["lights"], case["push"], mode="mix", radius_cutoff=bool(radius), want=("f64", "rgba8"))
    got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags)
    ref_nan = np.isnan(ref["f64"][..., :3]).any(-1)
    got_nan = np.isnan(got[..., :3]).any(-1)
    if radius != 30.0:  # every light takes part: the oracle agrees these are NaN (with a short radius only where one is near)
        assert ref_nan[2, 5:9].all() and ref_nan[3, 7] and ref_nan[7, 40:44].all()
    assert np.array_equal(got_nan, ref_nan)
    assert np.isnan(got[ref_nan][:, :3]).all()                             # all three channels
    assert_color_close(got[..., :3], ref["f64"][..., :3], ref["near_threshold"].astype(bool) | ref_nan, "finite pixels")
    got8, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R8G8B8A8_UNORM, flags=flags)
    assert (got8[ref_nan][:, :3] == 0).all() and (got8[..., 3] == 255).all()
    assert np.abs(got8.astype(int) - ref["rgba8"].astype(int)).max() <= 1
    # without lights nothing is evaluated on the normal: the ambient colour survives
    case0 = dict(case, lights=np.zeros((0, 12), np.float32))
    got0, _ = run_deferred_gpu(vkr, ctx, case0, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=flags)
    assert np.isfinite(got0).all() and np.allclose(got0[7, 40:44, :3], 0.05 * 200 / 255, rtol=1e-6)


def test_non_finite_lights(vkr, synth, oracle, ctx):
    """Light records with NaN / Inf positions: NaN exactly where the restated arithmetic (min / max as
    minNum / maxNum; GLSL leaves them undefined for NaN operands) yields NaN, i.e. on the pixels that face
    the viewer; back-facing pixels take the early-out and stay finite."""
    w, h, L = 48, 20, 6
    case = synth.deferred_case(w, h, L, config_index=12)
    good = case["lights"].copy().reshape(-1, 12)
    extra = np.zeros((3, 12), np.float32)
    extra[0, 0:3], extra[0, 4:7] = (np.nan, 0, 0), (5, 5, 5)
    extra[1, 0:3], extra[1, 4:7] = (np.inf, 1, 2), (5, 5, 5)
    extra[2, 0:3], extra[2, 4:7] = (1, -np.inf, 2), (5, 5, 5)
    for bad in (extra[:1], extra[1:2], extra[2:], extra):
        lights = np.concatenate([good[:3], bad, good[3:]])
        got, _ = run_deferred_gpu(vkr, ctx, dict(case, lights=lights), vkr.FORMAT_R32G32B32A32_SFLOAT)
        ref = oracle.deferred(case["gbuf"], lights, case["push"], mode="mix", want=("f64", "rgba8"))
        ref_nan = np.isnan(ref["f64"][..., :3]).any(-1)
        near = ref["near_threshold"].astype(bool)
        got_nan = np.isnan(got[..., :3]).any(-1)
        assert 0.5 < ref_nan.mean() < 1.0
        assert np.array_equal(got_nan | near, ref_nan | near)
        assert_color_close(got[..., :3], ref["f64"][..., :3], near | ref_nan, "pixels the bad light does not reach")
        got8, _ = run_deferred_gpu(vkr, ctx, dict(case, lights=lights), vkr.FORMAT_R8G8B8A8_UNORM)
        d = np.abs(got8.astype(int) - ref["rgba8"].astype(int))
        d[near] = 0
        assert d.max() <= 1 and (got8[ref_nan & ~near][:, :3] == 0).all()


@pytest.mark.parametrize("w,h,lights", [(96, 40, 24), (96, 40, 31), (96, 40, 33), (70, 50, 255), (131, 77, 256), (64, 36, 257), (50, 40, 600),
                                        (1366, 777, 40), (1290, 730, 300)])  # the last two: 32 x 32 tiles, cut off at the right and at the bottom
def test_lists_per_pixel_give_the_bytes_of_the_loop_over_every_light(vkr, synth, ctx, w, h, lights):
    """From 24 lights up the whole-table resolve gives every pixel a list of the lights that can contribute to it (a
    one-sided test of l.n against a margin, deferred.comp:43,84) and walks only those; VKR_DEFERRED_WHOLE_TILE_WORKGROUPS
    selects the loop that evaluates every light on every pixel.  Same bytes, in every format - on lights placed to
    sit on, just above and just below the tangent planes of pixels of the frame (where the test has to be one-sided),
    on lights beyond the test's range (1e15 and further), with colours that are not finite, and on a band of a taller
    frame (row_offset / full_height)."""
    case = synth.deferred_case(w, h, lights, config_index=9)
    g, push = case["gbuf"], case["push"]
    pos = synth.unproject(push[4:20], w, h, g["depth"])  # (h, w, 3), binary64
    n = g["normal"][..., :3].astype(np.float64)
    n = -n / np.linalg.norm(n, axis=-1, keepdims=True)
    rng = np.random.default_rng(lights)
    rec = np.ascontiguousarray(case["lights"], np.float32).reshape(-1, 12).copy()
    # every second light onto the tangent plane of a pixel of its own, displaced along the normal by 0, +-1e-7, +-1e-5,
    # +-1e-3 of the light's distance from the origin (the binary32 rounding of its position does the rest)
    steps = np.array([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3])
    for j in range(0, lights, 2):
        y, x = int(rng.integers(h)), int(rng.integers(w))
        t = np.cross(n[y, x], rng.standard_normal(3))
        t /= np.linalg.norm(t)
        q = pos[y, x] + t * rng.uniform(0.5, 60.0)
        q = q + n[y, x] * steps[(j // 2) % len(steps)] * np.abs(q).sum()
        rec[j, 0:3] = q.astype(np.float32)
    if lights >= 24:
        rec[5, 0:3] = (3e15, -1.0, 2.0)      # beyond the range of the test: every pixel walks it
        rec[7, 0:3] = (1.0, 2e20, -4e22)     # far enough for the squared distance to overflow
        rec[9, 4:7] = (np.inf, 1.0, 1.0)     # poisons what it reaches, on either path
        rec[11, 4] = np.nan
        rec[13, 1] = np.nan                  # positions that are not numbers
        rec[15, 0:3] = (np.inf, 0.0, -np.inf)
    g = {k: v.copy() for k, v in g.items()}
    g["normal"][3, 10:14, :3] = 0.0          # cleared texels (normalize(0)), a NaN normal, depths that are not numbers
    g["normal"][4, 20, 1] = np.float16(np.nan)
    g["depth"][5, 30] = np.nan
    g["depth"][6, 31] = np.inf
    g["depth"][7, 32:40] = 1.0               # the far plane: positions 10 000 units out
    case = dict(case, gbuf=g, lights=rec)
    for fmt in (vkr.FORMAT_R32G32B32A32_SFLOAT, vkr.FORMAT_R16G16B16A16_SFLOAT, vkr.FORMAT_R8G8B8A8_UNORM, vkr.FORMAT_B8G8R8A8_UNORM):
        lists, _ = run_deferred_gpu(vkr, ctx, case, fmt)
        loop, _ = run_deferred_gpu(vkr, ctx, case, fmt, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS)
        assert np.array_equal(lists.view(np.uint8), loop.view(np.uint8)), fmt
    rows = (h // 3, h // 3 + 17)
    lists, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, rows=rows)
    loop, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS, rows=rows)
    assert np.array_equal(lists.view(np.uint8), loop.view(np.uint8))
    if lights >= 24:
        full, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
        assert np.isnan(full[..., :3]).any() and np.isfinite(full[..., :3]).any()  # the bad colours reach some pixels, not all
        assert np.isnan(full[3, 10:14, :3]).all()  # cleared texels: normalize(0)
