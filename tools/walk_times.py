#!/usr/bin/env python3
"""Timeline of one launch of the per-pixel-list resolve (timing build: deferred.hip with -DVKR_X_TIMES, library path
in VKR_LIB_PATH): when its waves start, have their lists, and end; how many are alive over time; how long the launch
takes from its first wave's start to its last wave's end, beside the HIP-event time of the same launch.

    python tools/walk_times.py [rows=270] [lights=256]
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
    lib = vkr.lib()
    lib.vkr_x_wave_times.argtypes = [C.c_void_p, C.c_uint]
    ctx = vkr.Context(0)
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 270
    W, H, L = 3840, 2160, int(sys.argv[2]) if len(sys.argv) > 2 else 256
    case = synth.deferred_case(W, H, L, config_index=1)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    img = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    a = vkr.make_deferred_args(gb, img, lights, L, case["push"], row0=(540 if rows <= 1620 else 0), rows=rows)
    for _ in range(20):
        ctx.deferred_resolve(a)
    ctx.wait_idle()
    marks = [ctx.marker()]
    for _ in range(11):
        ctx.deferred_resolve(a)
        marks.append(ctx.marker())
    ctx.wait_idle()
    event_ms = sorted(marks[i + 1].elapsed_ms_since(marks[i]) for i in range(11))[5]
    tiles32 = ((rows + 31) // 32) * (W // 32)
    res = {"rows": rows, "lights": L, "event_ms_median": event_ms}
    # (the launcher's rule: 32 x 16 pixel tiles when 32 x 32 ones would make fewer than 16 workgroups per CU)
    shapes = ((16, 4),) if tiles32 < 16 * 256 else ((32, 8),)
    for tile_rows, waves_per_wg in shapes:
        wgs = ((rows + tile_rows - 1) // tile_rows) * (W // 32)
        waves = min(wgs * waves_per_wg, 65536)
        buf = np.zeros((waves, 4), np.uint64)
        assert lib.vkr_x_wave_times(buf.ctypes.data_as(C.c_void_p), waves) == 0
        start, end, lists = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 3].astype(np.int64)
        if not (start > 0).all() or not ((end - start) > 0).all() or (end.max() - start.min()) > 100 * 1e6:
            continue  # not the shape this launch used
        t0 = start.min()
        us = lambda x: (x - t0) / 100.0  # the wall clock counts at 100 MHz
        span = float(us(end).max())
        step = max(span / 40.0, 0.01)
        alive = [int(((us(start) <= k * step) & (us(end) > k * step)).sum()) for k in range(41)]
        wg_start = us(start).reshape(-1, waves_per_wg).min(1)
        wg_end = us(end).reshape(-1, waves_per_wg).max(1)
        first_round = wg_start < 0.25 * span
        q = lambda v: {"p10": float(np.percentile(v, 10)), "p50": float(np.median(v)), "p90": float(np.percentile(v, 90)), "max": float(v.max())}
        res.update({
            "tile_rows": tile_rows, "workgroups": int(wgs), "waves_recorded": int(waves),
            "first_start_to_last_end_us": span,
            "waves_alive_every_%.1f_us" % step: alive,
            "wave_start_us": q(us(start)), "wave_end_us": q(us(end)),
            "lists_and_sort_us_per_wave": q((lists - start) / 100.0), "walk_us_per_wave": q((end - lists) / 100.0),
            "workgroup_lifetime_us": q(wg_end - wg_start),
            "workgroups_in_first_round": int(first_round.sum()),
            "first_round_lifetime_us": q((wg_end - wg_start)[first_round]),
            "later_rounds_lifetime_us": q((wg_end - wg_start)[~first_round]) if (~first_round).any() else None,
            "spread_of_wave_ends_within_a_workgroup_us": q(us(end).reshape(-1, waves_per_wg).max(1) - us(end).reshape(-1, waves_per_wg).min(1)),
        })
        break
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
