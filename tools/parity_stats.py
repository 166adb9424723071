#!/usr/bin/env python3
"""Distance of the whole-table resolve (rgba32f) from the oracle's binary64-lighting evaluation: how many channels lie
beyond 1e-4 / 5e-5 / 2e-5 (relative, with the tests' absolute floor) and the largest distance, on the reference-sized
frame (1920x1080, 32 lights) and on a band of the headline frame (3840x2160, 256 lights).  For comparing builds:

    VKR_LIB_PATH=path/to/lib.so python tools/parity_stats.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import oracle
    from _util import rel_err, run_deferred_gpu
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    oracle.build()
    ctx = vkr.Context(0)
    out = {}
    for name, (w, h, l, idx, rows) in {"1080p_32": (1920, 1080, 32, 0, (0, 1080)), "4k_256_rows_0_270": (3840, 2160, 256, 1, (0, 270))}.items():
        case = synth.deferred_case(w, h, l, config_index=idx)
        got, _ = run_deferred_gpu(vkr, ctx, case, vkr.FORMAT_R32G32B32A32_SFLOAT)
        ref = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="mix", rows=rows, threads=16, want=("f64",))
        e = rel_err(got[rows[0]:rows[1], :, :3], ref["f64"][rows[0]:rows[1], :, :3])
        keep = ~ref["near_threshold"][rows[0]:rows[1]].astype(bool)
        e = e[keep]
        out[name] = {"channels": int(e.size), "beyond_1e-4": int((e > 1e-4).sum()), "beyond_5e-5": int((e > 5e-5).sum()),
                     "beyond_2e-5": int((e > 2e-5).sum()), "beyond_1e-5": int((e > 1e-5).sum()), "largest": float(e.max()), "mean": float(e.mean())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
