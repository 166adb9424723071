#!/usr/bin/env python3
"""A/B timing of builds of the library on one GPU: the whole-table resolve at the headline size, one
270-row band of it, and the reference-sized 1080p x 32 frame.  Builds alternate (A B A B ...), one process each,
so that clock drift of the box falls on all of them alike.

    python tools/ab_deferred.py name=path/to/lib.so [name=path ...] [--rounds 3]
"""
import argparse
import importlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    ctx = vkr.Context(0)

    def timed(args, steps, warm):
        for _ in range(warm):
            ctx.deferred_resolve(args)
        ctx.wait_idle()
        marks = [ctx.marker()]
        for _ in range(steps):
            ctx.deferred_resolve(args)
            marks.append(ctx.marker())
        ctx.wait_idle()
        ms = sorted(marks[i + 1].elapsed_ms_since(marks[i]) for i in range(steps))
        return ms[len(ms) // 2]

    out = {}
    case = synth.deferred_case(3840, 2160, 256, config_index=1)
    gb = vkr.GBuffer(ctx, 3840, 2160).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    img = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 3840, 2160)
    out["whole_4k_256"] = timed(vkr.make_deferred_args(gb, img, lights, 256, case["push"]), 30, 15)
    out["band_270"] = timed(vkr.make_deferred_args(gb, img, lights, 256, case["push"], row0=270 * 3, rows=270), 40, 10)
    out["band_540"] = timed(vkr.make_deferred_args(gb, img, lights, 256, case["push"], row0=540, rows=540), 40, 10)
    case = synth.deferred_case(1920, 1080, 32, config_index=0)
    gb2 = vkr.GBuffer(ctx, 1920, 1080).upload(case["gbuf"])
    l2 = ctx.buffer_from(case["lights"])
    img2 = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 1920, 1080)
    out["1080p_32"] = timed(vkr.make_deferred_args(gb2, img2, l2, 32, case["push"]), 40, 10)
    print(json.dumps(out))


def main():
    if "--worker" in sys.argv:
        return worker()
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    libs = [s.split("=", 1) for s in a.libs]
    res = {n: [] for n, _ in libs}
    for _ in range(a.rounds):
        for n, path in libs:
            env = dict(os.environ, VKR_LIB_PATH=os.path.abspath(path))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker"], env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                print(n, "failed:", r.stderr[-2000:], file=sys.stderr)
                sys.exit(1)
            res[n].append(json.loads(r.stdout.strip().splitlines()[-1]))
    summary = {n: {k: min(x[k] for x in runs) for k in runs[0]} for n, runs in res.items()}
    print(json.dumps({"min_of_medians_ms": summary, "runs": res}))


if __name__ == "__main__":
    main()
