#!/usr/bin/env python3
"""How many (wave, light-pair) trips of the resolve's light loop the exact early-outs remove, measured with a counting
build of the library (deferred.hip compiled with -DVKR_X_COUNT, path in VKR_LIB_PATH):

    VKR_LIB_PATH=tools/ab/libvkr_count.so python tools/skip_stats.py

Prints one JSON object: per workload the trips the waves would walk, those not walked at all (no pixel of the wave can
be lit), and those cut short by the per-trip test (every light of the trip behind every surface of the wave).
"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    frame = importlib.import_module("vk-renderer_amd.frame")
    lib = vkr.lib()
    if not hasattr(lib, "vkr_x_trip_counts"):
        raise SystemExit("not a counting build: compile deferred.hip with -DVKR_X_COUNT and point VKR_LIB_PATH at the library")
    lib.vkr_x_trip_counts.argtypes = [C.POINTER(C.c_ulonglong * 3), C.c_int]
    ctx = vkr.Context(0)

    def counts(reset=True):
        ctx.wait_idle()
        out = (C.c_ulonglong * 3)()
        assert lib.vkr_x_trip_counts(C.byref(out), 1 if reset else 0) == 0
        walked, by_test, not_walked = (int(v) for v in out)
        return {"trips": walked, "not_walked": not_walked, "cut_short_by_test": by_test,
                "fraction_removed": (not_walked + by_test) / max(walked, 1)}

    res = {}
    W, H, L = 3840, 2160, 256
    case = synth.deferred_case(W, H, L, config_index=1)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    counts()
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out, lights, L, case["push"]))
    res["headline_4k_256_uncorrelated_normals"] = counts()

    cam = synth.reference_camera(W, H)
    spheres = ctx.buffer_from(synth.make_spheres(64))
    frame.gbuffer_fill(ctx, gb, spheres, frame.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], 64))
    al = synth.deferred_case(64, 36, L, config_index=5)
    albuf = ctx.buffer_from(al["lights"])
    push = vkr.deferred_push(cam["pos"], cam["inv_vp"])
    counts()
    ctx.deferred_resolve(vkr.make_deferred_args(gb, out, albuf, L, push))
    res["analytic_4k_256_whole_table"] = counts()

    for (Wt, Ht, Lt) in ((3840, 2160, 4096), (7680, 4320, 4096)):
        gbt = vkr.GBuffer(ctx, Wt, Ht)
        camt = synth.reference_camera(Wt, Ht)
        frame.gbuffer_fill(ctx, gbt, spheres, frame.make_gbuffer_scene(camt["vp"], camt["inv_vp"], camt["pos"], 64))
        c = synth.deferred_case(64, 36, Lt, config_index=3)
        pusht = c["push"].copy()
        pusht[3] = 25.0
        lb = ctx.buffer_from(c["lights"])
        outt = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, Wt, Ht)
        counts()
        ctx.deferred_resolve(vkr.make_deferred_args(gbt, outt, lb, Lt, pusht, flags=vkr.DEFERRED_RADIUS_CUTOFF))
        res[f"analytic_{Wt}x{Ht}_{Lt}_light_lists"] = counts()
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
