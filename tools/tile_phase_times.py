#!/usr/bin/env python3
"""Where a wave of tile_lists_kernel spends its time, per phase (timing build: deferred.hip with -DVKR_X_TIMES,
library path in VKR_LIB_PATH):  python tools/tile_phase_times.py [tiled workload name of bench.py]"""
import ctypes as C
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ("depth loads + range", "tile bounds", "tile list", "pixel order (deep)", "group ranges + boxes (deep)", "group lists / dark pixels", "barrier + queue push")


def main():
    import bench
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    frame = importlib.import_module("vk-renderer_amd.frame")
    lib = vkr.lib()
    lib.vkr_x_wave_times.argtypes = [C.c_void_p, C.c_uint]
    name = sys.argv[1] if len(sys.argv) > 1 else "tiled_resolve_8k_discontinuous"
    deep = len(sys.argv) > 2 and sys.argv[2] == "deep"  # library built with -DVKR_X_TIMES=2: the deep-tile workgroups instead
    ctx = vkr.Context(0)
    line, keep = bench.tiled_line(ctx, vkr, synth, frame, 6, 3, name, {})
    if deep:
        buf = np.zeros((8192 * 8, 4), np.uint64)
        assert lib.vkr_x_wave_times(buf.ctypes.data_as(C.c_void_p), 8192 * 8) == 0
        st = buf.reshape(8192, 4, 8).astype(np.int64)
        st = st[st[:, 0, 0] != 0]  # tiles below 8192 that were deep
        names = ("requests issued", "barrier: planes in LDS", "first group shaded", "second group shaded", "barrier: all groups done", "rows stored + barrier")
        d = np.diff(st[:, :, :7], axis=2) / 100.0
        print(json.dumps({"workload": name, "ms_per_call": line["ms_per_call"], "deep_tiles_sampled": int(st.shape[0]),
                          "workgroup_time_per_tile_us": float((st[:, 0, 6] - st[:, 0, 0]).mean() / 100.0),
                          "phase_mean_us_by_wave": {names[i]: [float(d[:, w, i].mean()) for w in range(4)] for i in range(6)}}, indent=1))
        return
    tiles = min(line["lists"]["tiles"], 32768)
    buf = np.zeros((tiles * 2, 4), np.uint64)
    assert lib.vkr_x_wave_times(buf.ctypes.data_as(C.c_void_p), tiles * 2) == 0
    st = buf.reshape(tiles, 8).astype(np.int64)
    last = np.maximum.accumulate(st, axis=1)  # a phase a tile did not enter keeps the previous stamp
    d = np.diff(last, axis=1) / 100.0  # us (the wall clock counts at 100 MHz)
    res = {"workload": name, "ms_per_call": line["ms_per_call"], "tiles": tiles, "lists": line["lists"],
           "wave_lifetime_us": {"mean": float((last[:, 7] - st[:, 0]).mean() / 100.0), "p90": float(np.percentile(last[:, 7] - st[:, 0], 90) / 100.0)},
           "launch_span_us": float((last[:, 7].max() - st[:, 0].min()) / 100.0),
           "phase_mean_us": {PHASES[i]: float(d[:, i].mean()) for i in range(7)}}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
