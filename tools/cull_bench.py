#!/usr/bin/env python3
"""Per-call time of the cull at the headline size, rotating over input sets larger than the Infinity Cache.

    python tools/cull_bench.py [--instances N --meshes M --steps K --flags 0,1]
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
    ap.add_argument("--instances", type=int, default=1 << 22)
    ap.add_argument("--meshes", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--flags", default="0,1")
    a = ap.parse_args()
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    ctx = vkr.Context(0)
    scenes = [synth.cull_case(a.instances, a.meshes, config_index=2 + s) for s in range(3)]
    css = [vkr.CullScene(ctx, sc["aabb_min"], sc["aabb_max"], sc["mesh_base"], sc["cmd_template"]) for sc in scenes]
    res = {}
    for flags in [int(f) for f in a.flags.split(",")]:
        args = [cs.args(scenes[0]["planes"], flags) for cs in css]
        for i in range(12):
            ctx.prep_indirect(args[i % 3])
        ctx.wait_idle()
        e0 = ctx.marker()
        for i in range(a.steps):
            ctx.prep_indirect(args[i % 3])
        e1 = ctx.marker()
        ctx.wait_idle()
        us = e1.elapsed_ms_since(e0) / a.steps * 1e3
        vis = int(css[0].results()["cmds"]["instanceCount"].sum())
        byt = a.instances * 32.0 + vis * 4.0 + a.meshes * 20.0
        res[f"flags_{flags}"] = {"us_per_call": us, "hbm_frac": byt / (us * 1e-6) / 8e12, "visible": vis}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
