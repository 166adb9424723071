#!/usr/bin/env python3
"""Soak run on the GPU: thousands of back-to-back resolves and culls of the headline sizes, with the outputs
hashed at intervals - every interval must give the same bytes (the resolve and the command buffers are
deterministic; the atomic cull's list order is not, so only its commands are hashed).

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
    frames, commands = set(), set()
    for r in range(a.rounds):
        for _ in range(a.per_round):
            ctx.deferred_resolve(args)
            cs.run(sc["planes"], vkr.CULL_DETERMINISTIC)
            cs.run(sc["planes"], 0)
        ctx.wait_idle()
        frames.add(hashlib.sha256(out.download(np.uint8, 4).tobytes()).hexdigest())
        commands.add(hashlib.sha256(cs.results()["cmds"].tobytes()).hexdigest())
        print(f"round {r}: {len(frames)} frame hash(es), {len(commands)} command hash(es)", flush=True)
    assert len(frames) == 1 and len(commands) == 1, "outputs changed between rounds"
    print(f"soak ok: {a.rounds * a.per_round} resolves, {2 * a.rounds * a.per_round} culls")


if __name__ == "__main__":
    main()
