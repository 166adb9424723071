"""Multi-GPU sharding of the two passes: one process per GPU, torch.distributed (RCCL on ROCm).

Neither pass exchanges data while it computes:
  * deferred resolve : pixels are independent.  Ranks take horizontal screen bands (or whole frames
    of a batch); the only collective is the gather of the finished bands.
  * cull             : instances are independent; ranks take contiguous instance ranges cut on mesh
    boundaries, so no mesh segment is split and every command / list slot has exactly one writer.
    The collective is the gather (sum) of the per-mesh command counts and of the list slices.

The partition helpers are pure functions of (size, world) so that every rank derives the same
layout without communicating.  `compute` callables are injected by the caller: the product passes
the HIP entry points; CPU tests pass a stand-in.
"""
from __future__ import annotations

from typing import Callable, Sequence

import numpy as np


def band_rows(height: int, world: int, rank: int) -> tuple[int, int]:
    """Rows [r0, r1) of rank's band: height split as evenly as possible, earlier ranks get the extra row."""
    base, extra = divmod(height, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def frame_slices(frame_count: int, world: int, rank: int) -> range:
    """Frame-batch parallelism: frame f goes to rank f % world."""
    return range(rank, frame_count, world)


def mesh_range_cuts(mesh_base: np.ndarray, instance_count: int, world: int) -> list[tuple[int, int, int, int]]:
    """Per-rank (mesh0, mesh1, inst0, inst1): contiguous mesh ranges whose instance ranges are as close
    to instance_count/world as mesh boundaries allow.  Every mesh belongs to exactly one rank."""
    mesh_base = np.asarray(mesh_base, np.int64)
    m = mesh_base.shape[0]
    cuts_m = [0]
    for r in range(1, world):
        target = r * instance_count // world
        # first mesh whose base is >= target (never moving backwards)
        k = int(np.searchsorted(mesh_base, target, side="left"))
        cuts_m.append(max(min(k, m), cuts_m[-1]))
    cuts_m.append(m)
    out = []
    for r in range(world):
        m0, m1 = cuts_m[r], cuts_m[r + 1]
        i0 = int(mesh_base[m0]) if m0 < m else instance_count
        i1 = int(mesh_base[m1]) if m1 < m else instance_count
        out.append((m0, m1, i0, i1))
    return out


def sharded_deferred(height: int, world: int, rank: int, compute_band: Callable[[int, int], np.ndarray],
                     all_gather: Callable[[np.ndarray], Sequence[np.ndarray]]) -> np.ndarray:
    """Shades this rank's band and gathers the frame.  compute_band(r0, r1) -> (r1-r0, W, C) array;
    all_gather(x) -> list of every rank's x (bands may differ by one row)."""
    r0, r1 = band_rows(height, world, rank)
    band = compute_band(r0, r1)
    parts = all_gather(band)
    return np.concatenate(list(parts), axis=0)


def sharded_cull(mesh_base: np.ndarray, instance_count: int, world: int, rank: int,
                 compute_range: Callable[[int, int, int, int], tuple[np.ndarray, np.ndarray]],
                 all_gather: Callable[[np.ndarray], Sequence[np.ndarray]]) -> tuple[np.ndarray, np.ndarray]:
    """Culls this rank's mesh range and assembles the full command buffer and visible list.
    compute_range(m0, m1, i0, i1) -> (cmds[m1-m0] structured, visible[i1-i0] u32 with global indices)."""
    cuts = mesh_range_cuts(mesh_base, instance_count, world)
    m0, m1, i0, i1 = cuts[rank]
    cmds, visible = compute_range(m0, m1, i0, i1)
    all_cmds = all_gather(np.ascontiguousarray(cmds).view(np.uint8).reshape(-1))
    all_vis = all_gather(np.ascontiguousarray(visible, np.uint32))
    cmd_bytes = np.concatenate([np.asarray(c, np.uint8) for c in all_cmds])
    return cmd_bytes.view(cmds.dtype), np.concatenate([np.asarray(v, np.uint32) for v in all_vis])


def torch_all_gather_numpy(x: np.ndarray) -> list[np.ndarray]:
    """all_gather of host arrays whose leading sizes may differ, over the default process group
    (gloo on CPU; for device tensors the bench gathers torch tensors directly)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    flat = torch.from_numpy(np.ascontiguousarray(x).view(np.uint8).reshape(-1).copy())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([flat.numel()], dtype=torch.int64))
    cap = int(max(int(s.item()) for s in sizes))
    padded = torch.zeros(cap, dtype=torch.uint8)
    padded[:flat.numel()] = flat
    bufs = [torch.zeros(cap, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(bufs, padded)
    tail = x.shape[1:]
    out = []
    for b, s in zip(bufs, sizes):
        a = b[:int(s.item())].numpy().view(x.dtype)
        out.append(a.reshape((-1,) + tail))
    return out
