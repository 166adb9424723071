#!/usr/bin/env python3
"""How the per-pixel light lists of the whole-table resolve come out on the headline frame (numpy, no GPU): the share of
(pixel, light) pairs that stay on a list, and how many trips of four lights a batch of 64 pixels takes when the
pixels of a tile are sorted by list length - against the trips that lists of the mean length would take.

    python tools/list_length_sim.py [--width 3840 --height 2160 --lights 256 --rows 64]
"""
import argparse
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=256)
    ap.add_argument("--rows", type=int, default=64, help="rows of the frame to look at (from the top; a multiple of 64)")
    a = ap.parse_args()
    synth = importlib.import_module("vk-renderer_amd.synth")
    case = synth.deferred_case(a.width, a.height, a.lights, config_index=1)
    g, push = case["gbuf"], case["push"].astype(np.float64)
    rows = np.arange(a.rows)
    pos = synth.unproject(case["push"][4:20], a.width, a.height, g["depth"])[rows]
    n = g["normal"][rows][..., :3].astype(np.float64)
    n = -n / np.linalg.norm(n, axis=-1, keepdims=True)
    view = push[:3] - pos
    front = (view * n).sum(-1) > 0.0
    lp = np.ascontiguousarray(case["lights"], np.float32).reshape(-1, 12)[:, :3].astype(np.float64)
    ln = np.einsum("rwc,lc->rwl", n, lp) - (pos * n).sum(-1)[..., None]
    on_list = (ln > 0.0) & front[..., None]
    count = on_list.sum(-1)
    out = {"pairs_on_a_list": float(on_list.mean()), "front_facing_pixels": float(front.mean()),
           "longest_list_of_64_neighbours_over_table": float(count.reshape(a.rows, -1, 64).max(-1).mean() / a.lights), "tiles": {}}
    for tile_rows in (16, 32, 64):
        trips = ideal = batches = 0
        for ty in range(a.rows // tile_rows):
            for tx in range(a.width // 32):
                c = np.sort(count[ty * tile_rows:(ty + 1) * tile_rows, tx * 32:(tx + 1) * 32].ravel()).reshape(-1, 64)
                trips += np.ceil(c.max(1) / 4).sum()
                ideal += c.mean(1).sum() / 4
                batches += c.shape[0]
        out["tiles"][f"32x{tile_rows}"] = {"trips_per_batch": trips / batches, "trips_if_every_list_had_the_mean_length": ideal / batches}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
