#!/usr/bin/env python3
"""Where a screen band's launch spends its time: the 4K frame's rows [r0, r0 + rows) at several light counts and
band heights, median of back-to-back launches (HIP events).  VKR_LIB_PATH selects the build."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    ctx = vkr.Context(0)
    W, H = 3840, 2160
    case = synth.deferred_case(W, H, 256, config_index=1)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    img = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)

    def timed(rows, L, steps=40, warm=10, flags=0):
        a = vkr.make_deferred_args(gb, img, lights, L, case["push"], row0=min(540, H - rows), rows=rows, flags=flags)
        for _ in range(warm):
            ctx.deferred_resolve(a)
        ctx.wait_idle()
        marks = [ctx.marker()]
        for _ in range(steps):
            ctx.deferred_resolve(a)
            marks.append(ctx.marker())
        ctx.wait_idle()
        ms = sorted(marks[i + 1].elapsed_ms_since(marks[i]) for i in range(steps))
        return ms[len(ms) // 2]

    timed(2160 - 540, 256, 10, 10)  # clocks up
    res = {"rows270": {L: timed(270, L) for L in (0, 2, 64, 128, 192, 256)},
           "L256": {r: timed(r, 256) for r in (8, 32, 64, 128, 264, 270, 272, 540, 1080)},
           "L256_whole_tile_workgroups": {r: timed(r, 256, flags=vkr.DEFERRED_WHOLE_TILE_WORKGROUPS) for r in (32, 256, 270, 540)}}
    # the same number of (pixel, light) evaluations as 270 rows x 256 lights, in waves of half / a quarter the length
    res["equal_work"] = {"270x256": timed(270, 256), "540x128": timed(540, 128), "1080x64": timed(1080, 64), "2160x32": timed(2160, 32),
                         "270x0": timed(270, 0), "540x0": timed(540, 0), "1080x0": timed(1080, 0), "2160x0": timed(2160, 0)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
