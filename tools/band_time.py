#!/usr/bin/env python3
"""Kernel time of one rank's screen band of the headline frame, for every rank of an N-way split, on ONE GPU:
the compute part of the band-sharded time budget (the gather is not run here).

    python tools/band_time.py [--width W --height H --lights L --ranks 2 4 8 --steps K]
"""
import argparse
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=256)
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--whole-tiles", action="store_true", help="force one 256-thread workgroup per 32 x 32 tile (VKR_DEFERRED_WHOLE_TILE_WORKGROUPS)")
    a = ap.parse_args()
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    vdist = importlib.import_module("vk-renderer_amd.dist")
    ctx = vkr.Context(0)
    case = synth.deferred_case(a.width, a.height, a.lights, config_index=1)
    gb = vkr.GBuffer(ctx, a.width, a.height).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, a.width, a.height)
    res = {"width": a.width, "height": a.height, "lights": a.lights, "whole_tiles": a.whole_tiles, "bands": {}}
    for world in a.ranks:
        times = []
        for rank in range(world):
            r0, r1 = vdist.band_rows(a.height, world, rank)
            args = vkr.make_deferred_args(gb, out, lights, a.lights, case["push"], row0=r0, rows=r1 - r0,
                                          flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS if a.whole_tiles else 0)
            for _ in range(15 if not times else 3):
                ctx.deferred_resolve(args)
            ctx.wait_idle()
            e0 = ctx.marker()
            for _ in range(a.steps):
                ctx.deferred_resolve(args)
            e1 = ctx.marker()
            ctx.wait_idle()
            times.append(e1.elapsed_ms_since(e0) / a.steps)
        res["bands"][str(world)] = {"rows": vdist.band_rows(a.height, world, 0)[1], "ms_per_band": times, "max_ms": max(times)}
    whole = res["bands"].get("1", {}).get("max_ms")
    if whole:
        for world, b in res["bands"].items():
            b["compute_speedup"] = whole / b["max_ms"]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
