#!/usr/bin/env python3
"""When the waves of one band launch of the whole-table resolve start and end, and how the waves that share a SIMD
finish relative to each other (timing build: deferred.hip with -DVKR_X_TIMES, library path in VKR_LIB_PATH)."""
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
    W, H, L = 3840, 2160, int(sys.argv[2]) if len(sys.argv) > 2 else 256
    case = synth.deferred_case(W, H, L, config_index=1)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    img = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 270
    a = vkr.make_deferred_args(gb, img, lights, L, case["push"], row0=(540 if rows <= 1620 else 0), rows=rows)
    for _ in range(20):
        ctx.deferred_resolve(a)
    ctx.wait_idle()
    # the build under test decides the shape: single-wave workgroups of 64 pixels of a row for short launches (they
    # do not record), four-wave workgroups over 32 x 32 tiles otherwise
    waves = ((rows + 31) // 32) * (W // 32) * 4
    buf = np.zeros((waves, 4), np.uint64)
    assert lib.vkr_x_wave_times(buf.ctypes.data_as(C.c_void_p), waves) == 0
    start, end, where, cycles = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64), buf[:, 2], buf[:, 3].astype(np.int64)
    t0 = start.min()
    us = lambda x: (x - t0) / 100.0  # the wall clock counts at 100 MHz
    hw = (where & np.uint64(0xFFFFFFFF)).astype(np.int64)
    xcc = (where >> np.uint64(32)).astype(np.int64) & 0xF
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    alive = [(int(((start <= t0 + k * 100) & (end > t0 + k * 100)).sum())) for k in range(0, int((end.max() - t0) / 100) + 1, max(1, int((end.max() - t0) / 100) // 40))]
    res = {"waves": waves, "lights": L, "waves_alive_over_time": alive, "start_us": {"p50": float(np.median(us(start))), "max": float(us(start).max())},
           "end_us": {"min": float(us(end).min()), "p10": float(np.percentile(us(end), 10)), "p50": float(np.median(us(end))),
                      "p90": float(np.percentile(us(end), 90)), "max": float(us(end).max())},
           "lifetime_us": {"min": float(((end - start) / 100.0).min()), "p50": float(np.median((end - start) / 100.0)), "max": float(((end - start) / 100.0).max())},
           "cycles_per_wave": {"p50": float(np.median(cycles)), "min": float(cycles.min()), "max": float(cycles.max())}}
    groups = {}
    for k, e in zip(key.tolist(), us(end).tolist()):
        groups.setdefault(k, []).append(e)
    sizes = np.array([len(v) for v in groups.values()])
    spread = np.array([max(v) - min(v) for v in groups.values() if len(v) > 1])
    last = np.array([max(v) for v in groups.values()])
    wid = hw & 15
    res["wave_id_histogram"] = {int(k): int((wid == k).sum()) for k in np.unique(wid)}
    ids = {}
    for k, w_ in zip(key.tolist(), wid.tolist()):
        ids.setdefault(k, []).append(w_ & 3)
    res["simds_with_distinct_low_id_bits"] = int(sum(1 for v in ids.values() if len(set(v)) == len(v)))
    res["simds_seen"] = len(groups)
    res["waves_per_simd_histogram"] = {int(k): int((sizes == k).sum()) for k in np.unique(sizes)}
    res["finish_spread_within_simd_us"] = {"p50": float(np.median(spread)), "p90": float(np.percentile(spread, 90)), "max": float(spread.max())}
    res["last_finish_per_simd_us"] = {"min": float(last.min()), "p50": float(np.median(last)), "max": float(last.max())}
    by_xcc = {}
    for x in np.unique(xcc):
        by_xcc[int(x)] = {"waves": int((xcc == x).sum()), "last_end_us": float(us(end)[xcc == x].max()), "median_end_us": float(np.median(us(end)[xcc == x]))}
    res["by_xcc"] = by_xcc
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
