#!/usr/bin/env python3
"""Two frames on which the reference's deferred shader is hard to reproduce in binary32, with the colours its own
text (/root/reference/shaders/deferred.comp, read at run time, evaluated by tests/glsl_eval.py) gives for them in
binary64 and, operation by operation, in binary32:

    python tests/golden/make_shader_eval_hard.py  ->  tests/golden/shader_text_mirror_peak_64x8_L13.npz
                                                      tests/golden/shader_text_near_light_32x8_L13.npz

mirror peak   a flat, smooth (roughness 0 ... just above the 0.0064 floor of alpha), mostly metallic wall seen
              head-on, with lights placed on the reflection of the view ray: along each row the half vector sweeps
              through the normal, where the GGX denominator (a2 - 1) * hDotN^2 + 1 is the difference of two numbers
              4e-5 apart and a literal binary32 evaluation loses three digits.
near light    surfaces 320-384 units from the camera with lights 0.05-0.5 units in front of them: 1 / distSqr
              reaches 400 and hinges on the last bits of the reconstructed position.

In both frames inverseVP has one power-of-two entry per row and column and w = 1, and the extent is a power of two,
so the position the shader reconstructs is exact in binary32 and in binary64 alike: what differs between a binary32
run and these colours is the lighting arithmetic alone, which is what the fixtures are for.  Fixtures are data: the
inputs and the colours; none of the shader's text is stored."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SHADER = "/root/reference/shaders/deferred.comp"


def push_for(cam, scale_xy, scale_z):
    """DeferredPushData: cameraPos, pointLightRadius, inverseVP (column-major) = diag(scale_xy, scale_xy, scale_z, 1)."""
    push = np.zeros(20, np.float32)
    push[0:3] = cam
    push[3] = 1e6
    m = np.zeros((4, 4), np.float32)  # m[c] = column c
    m[0, 0] = m[1, 1] = scale_xy
    m[2, 2] = scale_z
    m[3, 3] = 1.0
    push[4:20] = m.reshape(16)
    return push


def positions(push, w, h, depth):
    x = ((np.arange(w) + 0.5) / w * 2.0 - 1.0)[None, :] * float(push[4])
    y = ((np.arange(h) + 0.5) / h * 2.0 - 1.0)[:, None] * float(push[9])
    return np.stack([np.broadcast_to(x, (h, w)), np.broadcast_to(y, (h, w)), depth.astype(np.float64) * float(push[14])], -1)


def mirror_peak():
    w, h = 64, 8
    rng = np.random.default_rng(116)
    push = push_for((0.0, 0.0, 0.0), 2.0, 1024.0)
    depth = np.full((h, w), 0.125, np.float32)  # a wall at z = 128
    P = positions(push, w, h, depth)
    stored = np.zeros((h, w, 4), np.float16)  # the shader shades with -normalize(stored.xyz)
    stored[..., 2] = 1.0
    stored[..., 3] = 1.0
    stored[4, :, 0:3] = np.array([0.03125, -0.0625, 1.0], np.float16) * np.float16(0.5)
    tilt = rng.integers(-3, 4, (3, w, 2)) / 64.0
    stored[5:8, :, 0:2] = tilt.astype(np.float16)
    albedo = np.full((h, w, 4), 255, np.uint8)
    albedo[..., :3] = rng.integers(32, 256, (h, w, 3))
    material = np.full((h, w, 4), 255, np.uint8)
    material[..., 1] = 0          # roughness 0: alpha at its floor
    material[1, :, 1] = 20        # roughness^2 = 0.00615: still the floor
    material[2, :, 1] = 21        # 0.00678: just above it
    material[3, :, 2] = 0         # a dielectric row: F = (1 - vDotH)^5
    material[4, :, 2] = 128
    material[5:8, :, 1] = rng.integers(0, 31, (3, w))
    material[5:8, :, 2] = rng.integers(0, 256, (3, w))
    lights = np.zeros((13, 12), np.float32)  # the light count the reference's text is compiled with

    def mirrored(px, py, along):  # a light on the reflection of pixel (px, py)'s view ray about its normal
        p = P[py, px]
        n = -stored[py, px, :3].astype(np.float64)
        n /= np.linalg.norm(n)
        v = -p / np.linalg.norm(p)
        return p + along * (2.0 * (v @ n) * n - v)

    spots = [(32, 0, 64.0), (11, 2, 30.0), (40, 4, 100.0), (50, 6, 12.0), (20, 1, 90.0), (58, 3, 45.0), (5, 5, 70.0), (29, 7, 20.0)]
    for i, (px, py, along) in enumerate(spots):
        lights[i, 0:3] = mirrored(px, py, along)
    lights[8, 0:3] = (3.0, -2.0, -200.0)  # behind the camera, far: a small even term
    lights[9, 0:3] = (-1.5, 1.0, 100.0)
    lights[10, 0:3] = (0.75, 0.5, 120.0)
    lights[11, 0:3] = (40.0, -30.0, 60.0)  # grazing
    lights[12, 0:3] = (0.0, 0.0, 127.0)    # one unit in front of the wall
    lights[:, 4:7] = rng.uniform(10.0, 60.0, (13, 3))
    lights[12, 4:7] *= 0.01
    return "shader_text_mirror_peak_64x8_L13.npz", w, h, {"albedo": albedo, "material": material, "normal": stored, "depth": depth}, lights, push


def near_light():
    w, h = 32, 8
    rng = np.random.default_rng(121)
    push = push_for((0.5, -0.25, 0.0), 16.0, 1024.0)
    depth = (0.3125 + rng.integers(0, 256, (h, w)) * 2.0 ** -12).astype(np.float32)  # z = 320 ... 383.75 in steps of 0.25
    P = positions(push, w, h, depth)
    cam = push[0:3].astype(np.float64)
    nrm = rng.normal(size=(h, w, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    facing = np.sum(-nrm * (cam - P), -1) > 0  # the shader negates the stored normal
    nrm[~facing] *= -1.0
    nrm *= rng.uniform(0.5, 2.0, (h, w, 1))
    stored = np.ones((h, w, 4), np.float16)
    stored[..., :3] = nrm.astype(np.float16)
    albedo = np.full((h, w, 4), 255, np.uint8)
    albedo[..., :3] = rng.integers(16, 256, (h, w, 3))
    material = np.full((h, w, 4), 255, np.uint8)
    material[..., 1] = rng.integers(0, 256, (h, w))
    material[..., 2] = np.where(rng.uniform(size=(h, w)) < 0.5, 0, rng.integers(0, 256, (h, w)))
    lights = np.zeros((13, 12), np.float32)
    spots = [(3, 1), (9, 6), (14, 2), (20, 5), (27, 1), (30, 7), (6, 4), (17, 7), (23, 3), (12, 0)]
    for i, (px, py) in enumerate(spots):
        n = -stored[py, px, :3].astype(np.float64)
        n /= np.linalg.norm(n)
        side = rng.normal(size=3)
        side -= (side @ n) * n
        side /= np.linalg.norm(side)
        dist = 0.05 + 0.45 * i / (len(spots) - 1)
        lights[i, 0:3] = P[py, px] + dist * (0.8 * n + 0.6 * side)  # in front of the surface, off its normal
        lights[i, 4:7] = rng.uniform(0.01, 0.05, 3)
    lights[10, 0:3], lights[10, 4:7] = (0.0, 0.0, 200.0), (3000.0, 2500.0, 2000.0)
    lights[11, 0:3], lights[11, 4:7] = (-12.0, 5.0, 340.0), (40.0, 80.0, 60.0)
    lights[12, 0:3], lights[12, 4:7] = (9.0, -3.0, 100.0), (500.0, 100.0, 900.0)
    return "shader_text_near_light_32x8_L13.npz", w, h, {"albedo": albedo, "material": material, "normal": stored, "depth": depth}, lights, push


def main():
    import glsl_eval
    from glsl_eval import run_deferred_shader
    import importlib
    oracle = importlib.import_module("oracle")
    text = open(SHADER).read()
    manifest_path = os.path.join(HERE, "MANIFEST.json")
    manifest = json.load(open(manifest_path))
    for name, w, h, gbuf, lights, push in (mirror_peak(), near_light()):
        # exact reconstruction: the binary32 restatement's positions are the binary64 ones
        near = oracle.deferred(gbuf, lights, push, mode="mix", want=("f64",))["near_threshold"]
        assert not near.any(), f"{name}: a light sits on the 1e-6 early-out of {int(near.sum())} pixel(s); change the seed"
        out = run_deferred_shader(text, gbuf, lights, push, w, h)
        glsl_eval.set_precision(np.float32)
        try:
            with np.errstate(all="ignore"):
                out32 = run_deferred_shader(text, gbuf, lights, push, w, h).astype(np.float32)
        finally:
            glsl_eval.set_precision(np.float64)
        path = os.path.join(HERE, name)
        np.savez_compressed(path, albedo=gbuf["albedo"], material=gbuf["material"], normal=gbuf["normal"], depth=gbuf["depth"],
                            lights=lights, push=push, color=out, color_f32=out32)
        z = np.load(path)
        manifest[name] = {k: hashlib.sha256(np.ascontiguousarray(z[k]).tobytes()).hexdigest() for k in sorted(z.files)}
        rel = np.abs(out32[..., :3].astype(np.float64) - out[..., :3]) / (np.abs(out[..., :3]) + 2e-2)
        print(name, out.shape, "mean", float(out[..., :3].mean()), "max", float(out[..., :3].max()),
              "binary32 text vs binary64 text: max rel", float(rel.max()), "channels > 1e-4:", float((rel > 1e-4).mean()))
    json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
