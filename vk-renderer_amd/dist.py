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


# ----------------------------------------------------------------------------------------------- pipelines
# One frame (or one scene) split over the ranks, its pieces gathered on rank 0, step after step.  The piece
# buffers are double-buffered and the gather is asynchronous: the gather of step k runs while step k+1 is
# computed, and a buffer is written again only after the gather that last read it has been waited for.
# How buffers are allocated, gathered and waited for is the transport's business (RCCL on device tensors
# ordered through the library's stream, or gloo through host memory); what fills a buffer is injected by the
# caller (the HIP entry points in production, a stand-in in CPU tests).

class HostTransport:
    """gloo through host memory.  `device`: where the send buffers live ("cpu", or "cuda" when a kernel fills
    them and the gather is a rehearsal on a box with fewer GPUs than ranks); `sync`: called before a send
    buffer is copied to the host (the library's wait_idle), None for CPU buffers."""

    name = "gloo"

    def __init__(self, device: str = "cpu", sync: Callable[[], None] | None = None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.device, self.sync = device, sync
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def alloc_send(self, nbytes: int):
        return self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.device)

    def alloc_recv(self, nbytes: int):
        return self.torch.zeros(nbytes, dtype=self.torch.uint8)

    def gather_async(self, send, recv_list):
        if self.sync is not None:
            self.sync()  # the copy below is not ordered with the library's stream
        host = send.cpu() if send.device.type != "cpu" else send
        if self.world == 1:
            recv_list[0].copy_(host)
            return None
        return self.dist.gather(host, recv_list if self.rank == 0 else None, dst=0, async_op=True)

    def wait(self, work):
        if work is not None:
            work.wait()

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.world == 1:
            return value
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def to_numpy(self, tensor) -> np.ndarray:
        return tensor.cpu().numpy()

    def settle(self):
        """Buffers just allocated are zero-filled on torch's stream: finished before anything else writes them."""
        if self.device != "cpu":
            self.torch.cuda.synchronize()


class RcclTransport:
    """RCCL on device tensors.  torch sees the library's stream as an external stream: a collective issued
    while it is current is ordered after the kernels already on it, and waiting for the collective's work
    handle makes that stream (not the host) wait for the collective.  Under RCCL a gather is a group of
    point-to-point transfers: rank 0 receives from each peer over that peer's own xGMI link."""

    name = "nccl"

    def __init__(self, hip_stream: int, device_index: int):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = torch.device("cuda", device_index)
        self.stream = torch.cuda.ExternalStream(hip_stream, device=self.device)

    def alloc_send(self, nbytes: int):
        return self.torch.zeros(nbytes, dtype=self.torch.uint8, device=self.device)

    alloc_recv = alloc_send

    def gather_async(self, send, recv_list):
        with self.torch.cuda.stream(self.stream):
            return self.dist.gather(send, recv_list if self.rank == 0 else None, dst=0, async_op=True)

    def wait(self, work):
        if work is not None:
            with self.torch.cuda.stream(self.stream):
                work.wait()  # stream-level: later kernels on the library's stream start after the gather

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def to_numpy(self, tensor) -> np.ndarray:
        # the copy goes on the library's stream, behind whatever wait() put there: torch's own stream is ordered
        # neither with it (a non-blocking stream) nor with RCCL's
        with self.torch.cuda.stream(self.stream):
            return tensor.cpu().numpy()

    def settle(self):
        """Buffers just allocated are zero-filled on torch's stream, which is not ordered with the library's stream:
        finished before a kernel or a transfer writes them."""
        self.torch.cuda.synchronize(self.device)


class GatherPipeline:
    """`slots` send buffers of `nbytes` per rank and, on rank 0, `slots` sets of `world` receive buffers.
    step(fill): waits for the gather that last read the next buffer, calls fill(buffer) (which enqueues or
    performs the work that writes it) and starts the buffer's gather.  drain(): waits for every gather."""

    def __init__(self, transport, nbytes: int, slots: int = 2):
        self.t = transport
        self.slots = slots
        self.send = [transport.alloc_send(nbytes) for _ in range(slots)]
        self.recv = [[transport.alloc_recv(nbytes) for _ in range(transport.world)] if transport.rank == 0 else None for _ in range(slots)]
        self.works = [None] * slots
        self.steps = 0
        transport.settle()

    def step(self, fill: Callable[[object], None]):
        b = self.steps % self.slots
        self.t.wait(self.works[b])  # the gather that last read this buffer
        self.works[b] = None
        fill(self.send[b])
        self.works[b] = self.t.gather_async(self.send[b], self.recv[b])
        self.steps += 1

    def drain(self):
        for b in range(self.slots):
            self.t.wait(self.works[b])
            self.works[b] = None

    def latest(self):
        """Rank 0: the receive buffers of the most recent step (call drain() first)."""
        assert self.steps > 0
        return self.recv[(self.steps - 1) % self.slots]


class BandShardedResolve:
    """ONE frame resolved in horizontal bands, one band per rank, the rgba8 bands gathered on rank 0.
    resolve_band(buffer, r0, r1) shades rows [r0, r1) of the full frame into `buffer` (row pitch `pitch`
    bytes, band-relative rows); in production it is vkr_deferred_resolve with row_offset / full_height and
    the output image pointing at the buffer."""

    def __init__(self, transport, width: int, height: int, pitch: int, resolve_band: Callable[[object, int, int], None], slots: int = 2):
        self.t, self.width, self.height, self.pitch = transport, width, height, pitch
        self.r0, self.r1 = band_rows(height, transport.world, transport.rank)
        self.rows_max = band_rows(height, transport.world, 0)[1]
        self.resolve_band = resolve_band
        self.pipe = GatherPipeline(transport, max(self.rows_max * pitch, 16), slots)

    def step(self):
        self.pipe.step(lambda buf: self.resolve_band(buf, self.r0, self.r1))

    def drain(self):
        self.pipe.drain()

    def frame(self) -> np.ndarray | None:
        """Rank 0: the most recent frame, (height, width, 4) uint8, assembled from the gathered bands."""
        self.pipe.drain()
        if self.t.rank != 0:
            return None
        parts = []
        for r, buf in enumerate(self.pipe.latest()):
            a, b = band_rows(self.height, self.t.world, r)
            parts.append(self.t.to_numpy(buf).reshape(self.rows_max, self.pitch)[: b - a, : self.width * 4])
        return np.concatenate(parts).reshape(self.height, self.width, 4)


class RangeShardedCull:
    """ONE scene culled in instance ranges cut on mesh boundaries, one range per rank.  The only exchange is the
    gather of the per-mesh commands (20 bytes per mesh) to rank 0; the compacted lists stay on the rank that
    made them.  cull_range(buffer, m0, m1, i0, i1) culls meshes [m0, m1) = instances [i0, i1) and leaves their
    commands at the start of `buffer` (production: vkr_prep_indirect with first_mesh / first_instance and
    cmd_out pointing at the buffer)."""

    CMD_BYTES = 20

    def __init__(self, transport, mesh_base: np.ndarray, instance_count: int, cull_range: Callable[[object, int, int, int, int], None], slots: int = 2):
        self.t = transport
        self.cuts = mesh_range_cuts(mesh_base, instance_count, transport.world)
        self.mine = self.cuts[transport.rank]
        self.cull_range = cull_range
        cap = max(max(c[1] - c[0] for c in self.cuts) * self.CMD_BYTES, 16)
        self.pipe = GatherPipeline(transport, cap, slots)

    def step(self):
        self.pipe.step(lambda buf: self.cull_range(buf, *self.mine))

    def drain(self):
        self.pipe.drain()

    def commands(self) -> bytes | None:
        """Rank 0: the command buffer of the whole scene (mesh order), from the most recent step."""
        self.pipe.drain()
        if self.t.rank != 0:
            return None
        return b"".join(self.t.to_numpy(buf)[: (c[1] - c[0]) * self.CMD_BYTES].tobytes() for buf, c in zip(self.pipe.latest(), self.cuts))


def timed_steps(transport, pipeline, steps: int, warmup: int, device_sync: Callable[[], None]) -> float:
    """Seconds (max over ranks) for `steps` steps of a pipeline after `warmup` untimed ones, every gather
    finished on both sides of the timed region."""
    import time
    for _ in range(warmup):
        pipeline.step()
    pipeline.drain()
    device_sync()
    transport.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        pipeline.step()
    pipeline.drain()
    device_sync()
    transport.barrier()
    return transport.max_over_ranks(time.perf_counter() - t0)


def hip_band_resolver(vkr, ctx, gbuf, out_img, lights_buf, light_count: int, push, flags: int = 0):
    """resolve_band of BandShardedResolve on the HIP path: vkr_deferred_resolve of rows [r0, r1) with the output
    image placed in the transport's buffer (rgba8, the pitch of `out_img`)."""
    cache = {}

    def resolve_band(buf, r0, r1):
        a = cache.get(buf.data_ptr())
        if a is None:
            a = vkr.make_deferred_args(gbuf, out_img, lights_buf, light_count, push, flags=flags, row0=r0, rows=r1 - r0)
            a.out = vkr.Image2D(buf.data_ptr(), gbuf.width, r1 - r0, out_img.pitch, vkr.FORMAT_R8G8B8A8_UNORM)
            cache[buf.data_ptr()] = a
        ctx.deferred_resolve(a)

    return resolve_band


def hip_range_culler(vkr, ctx, scene: dict, planes, flags: int = 0):
    """cull_range of RangeShardedCull on the HIP path: this rank's slice of `scene` (host arrays aabb_min, aabb_max,
    mesh_base, cmd_template) is uploaded once; each call is vkr_prep_indirect with first_mesh / first_instance set and
    cmd_out pointing at the transport's buffer."""
    state = {}

    def cull_range(buf, m0, m1, i0, i1):
        part = state.get("part")
        if part is None:
            part = vkr.CullScene(ctx, scene["aabb_min"][i0:i1], scene["aabb_max"][i0:i1], scene["mesh_base"][m0:m1] - np.uint32(i0),
                                 scene["cmd_template"][m0:m1], first_mesh=m0, first_instance=i0)
            state["part"] = part
        a = state.get(buf.data_ptr())
        if a is None:
            a = part.args(planes, flags)
            a.cmd_out = buf.data_ptr()  # the commands of this rank's meshes land in the buffer that is gathered
            state[buf.data_ptr()] = a
        ctx.prep_indirect(a)

    return cull_range
