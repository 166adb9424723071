"""Shared helpers for the parity tests."""
from __future__ import annotations

import ctypes as C

import numpy as np

# north_star tolerance for the shaded colour: 1e-4 relative (max-abs), <= 1 LSB after the 8-bit store.
REL_TOL = 1e-4
# Absolute floor far below one 8-bit LSB (3.9e-3): a light whose l.n sits within rounding of the 1e-6
# early-out threshold contributes O(1e-6) or nothing, legitimately either way.
ABS_TOL = 2e-6


def rel_err(got: np.ndarray, ref: np.ndarray) -> np.ndarray:
    got = got.astype(np.float64)
    ref = ref.astype(np.float64)
    return np.abs(got - ref) / (np.abs(ref) + ABS_TOL / REL_TOL)


# How many binary32 ulps of the unit view / light directions a channel may additionally move by (see
# `view_sensitivity`): the kernels keep the light vector's frame components, 1/dist and the view vector's components in
# binary32, each good to about an ulp.
DIRECTION_ULPS = 2.0


def assert_color_close(got_rgb: np.ndarray, ref_rgb: np.ndarray, skip_mask: np.ndarray | None = None, what: str = "",
                       view_sensitivity: np.ndarray | None = None):
    """|got - ref| <= REL_TOL*|ref| + ABS_TOL on every channel of every pixel not in skip_mask.
    view_sensitivity (the oracle's conditioning probe: how far a channel moves when the unit view vector moves by one
    binary32 ulp across itself) widens the bound by DIRECTION_ULPS times itself.  It is of the order of 1e-7 of the
    colour except at GGX peaks of near-mirror surfaces, where no evaluation that keeps its unit vectors in binary32
    (the reference's included) resolves the colour to 1e-4; callers that pass it also bound how many channels lie
    beyond the plain bound."""
    got = got_rgb.astype(np.float64)
    ref = ref_rgb.astype(np.float64)
    tol = REL_TOL * np.abs(ref) + ABS_TOL
    if view_sensitivity is not None:
        tol = tol + DIRECTION_ULPS * np.nan_to_num(view_sensitivity.astype(np.float64), nan=0.0, posinf=0.0)
    bad = np.abs(got - ref) > tol
    bad |= ~np.isfinite(got) & np.isfinite(ref)
    if skip_mask is not None:
        bad &= ~skip_mask.astype(bool)[..., None]
    if bad.any():
        idx = np.argwhere(bad)[0]
        y, x, c = idx
        raise AssertionError(f"{what}: {int(bad.sum())} channel(s) out of tolerance; first at (y={y},x={x},c={c}): "
                             f"got {got[y, x, c]!r} ref {ref[y, x, c]!r} rel {abs(got[y, x, c] - ref[y, x, c]) / max(abs(ref[y, x, c]), 1e-30):.3e}")


def run_deferred_gpu(vkr, ctx, case: dict, out_format: int, *, flags: int = 0, rows=None, pitch_pad: int = 0):
    """Uploads a synthetic case, runs the typed entry point, returns (array, objects)."""
    w, h = case["width"], case["height"]
    gb = vkr.GBuffer(ctx, w, h).upload(case["gbuf"])
    lights_np = np.ascontiguousarray(case["lights"], np.float32).reshape(-1, 12)
    lights = ctx.buffer_from(lights_np) if lights_np.size else ctx.buffer(64)
    out = ctx.image(out_format, w, h)
    out.buffer.fill_u32(0)
    r0, nr = (0, h) if rows is None else (rows[0], rows[1] - rows[0])
    args = vkr.make_deferred_args(gb, out, lights, lights_np.shape[0], case["push"], flags=flags, row0=r0, rows=nr)
    ctx.deferred_resolve(args)
    ctx.wait_idle()
    if out_format in (vkr.FORMAT_R8G8B8A8_UNORM, vkr.FORMAT_B8G8R8A8_UNORM):
        arr = out.download(np.uint8, 4)
    elif out_format == vkr.FORMAT_R16G16B16A16_SFLOAT:
        arr = out.download(np.float16, 4)
    else:
        arr = out.download(np.float32, 4)
    return arr, (gb, lights, out)


def sorted_segments(visible: np.ndarray, mesh_base: np.ndarray, counts: np.ndarray) -> list:
    return [np.sort(visible[int(b):int(b) + int(c)]) for b, c in zip(mesh_base, counts)]


def image_row(img, y: int, dtype, channels: int) -> np.ndarray:
    """Row y of a device image (one pitched row is downloaded, not the plane)."""
    raw = img.buffer.download(np.uint8, img.pitch, offset=y * img.pitch)[: img.width * img.texel_bytes]
    out = np.ascontiguousarray(raw).view(dtype)
    return out.reshape(img.width, channels) if channels > 1 else out.reshape(img.width)


def oracle_row(oracle, planes: dict, y: int, height: int, lights, push, *, radius_cutoff: bool = False, mode: str = "mix"):
    """The oracle's 8-bit result and near-threshold mask for row y of a frame `height` rows tall, given only that
    row's G-buffer texels (planes: 'albedo' (W,4) u8, 'material' (W,4) u8, 'normal' (W,4) f16, 'depth' (W,) f32).
    The frame-sized arrays the oracle takes are zero-pitch views of the one row, so an 8K frame costs no memory."""
    from numpy.lib.stride_tricks import as_strided
    g = {}
    for k, a in planes.items():
        a = np.ascontiguousarray(a)
        g[k] = as_strided(a, shape=(height,) + a.shape, strides=(0,) + a.strides, writeable=False)
    res = oracle.deferred(g, lights, push, mode=mode, radius_cutoff=radius_cutoff, rows=(y, y + 1), want=("rgba8",))
    return res["rgba8"][y], res["near_threshold"][y].astype(bool)
