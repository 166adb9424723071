#!/usr/bin/env python3
"""Soak run on the GPU: thousands of back-to-back resolves and culls of the headline sizes, with the outputs
hashed at intervals - every interval must give the same bytes (both resolves, the command buffers and the
deterministic culls' lists; the atomic cull's list order is not fixed, so it runs ahead of the deterministic ones).

    python tools/soak.py [--rounds 8 --per-round 250]
"""
import argparse
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--per-round", type=int, default=250)
    ap.add_argument("--culls-per-resolve", type=int, default=4)
    a = ap.parse_args()
    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    ctx = vkr.Context(0)
    W, H, L = 3840, 2160, 256
    case = synth.deferred_case(W, H, L, config_index=1)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    args = vkr.make_deferred_args(gb, out, lights, L, case["push"])
    sc = synth.cull_case(1 << 22, 1024, config_index=2)
    cs = vkr.CullScene(ctx, sc["aabb_min"], sc["aabb_max"], sc["mesh_base"], sc["cmd_template"])
    # the radius-cutoff resolve (light lists, alternating queue counters) on a frame of its own
    tcase = synth.deferred_case(1920, 1080, 2048, config_index=3)
    tpush = tcase["push"].copy()
    tpush[3] = 25.0
    tgb = vkr.GBuffer(ctx, 1920, 1080).upload(tcase["gbuf"])
    tlights = ctx.buffer_from(tcase["lights"])
    tout = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, 1920, 1080)
    targs = vkr.make_deferred_args(tgb, tout, tlights, 2048, tpush, flags=vkr.DEFERRED_RADIUS_CUTOFF)
    frames, commands, lists, tiled = set(), set(), set(), set()
    for r in range(a.rounds):
        for i in range(a.per_round):
            ctx.deferred_resolve(args)
            cs.run(sc["planes"], 0)
            for _ in range(a.culls_per_resolve):  # the single-launch cull, whose blocks wait on each other's words
                cs.run(sc["planes"], vkr.CULL_DETERMINISTIC)
                ctx.prep_indirect_noalloc(cs.args(sc["planes"], 0))
            ctx.deferred_resolve(targs)
        ctx.wait_idle()
        res = cs.results()
        assert res["error"] == 0, "a cull reported an error"
        frames.add(hashlib.sha256(out.download(np.uint8, 4).tobytes()).hexdigest())
        commands.add(hashlib.sha256(res["cmds"].tobytes()).hexdigest())
        lists.add(hashlib.sha256(res["visible"].tobytes()).hexdigest())  # last writer was a deterministic call
        tiled.add(hashlib.sha256(tout.download(np.uint8, 4).tobytes()).hexdigest())
        print(f"round {r}: {len(frames)} frame, {len(commands)} command, {len(lists)} list, {len(tiled)} light-list frame hash(es)", flush=True)
    assert len(frames) == 1 and len(commands) == 1 and len(lists) == 1 and len(tiled) == 1, "outputs changed between rounds"
    n = a.rounds * a.per_round
    print(f"soak ok: {2 * n} resolves, {n * (1 + 2 * a.culls_per_resolve)} culls")


if __name__ == "__main__":
    main()
