#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[1]): deferred lighting resolve of a 3840x2160 synthetic
G-buffer with 256 point lights, 8-bit output, inputs resident in HBM.  One "step" = one resolve of
one frame.  The 2^22-instance frustum cull (configs[2]) is timed in the same run and reported in
the "cull" object.  For N > 1 every rank shades its own frame (frame-batch parallelism, no
collective in the timed region: weak scaling); the screen-band variant of ONE frame, its bands
gathered on rank 0 over RCCL behind the next frame's resolve, is timed as well and reported under
"band_sharded" with the gathered frame verified against rank 0's own whole-frame resolve.
`python3 bench.py --gpus N` starts its own N ranks (one process per GPU, rendezvous on 127.0.0.1);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E (spec)
VALU_FP32_PEAK = 157.3e12  # FLOP/s, vector FP32 (spec)


# Kernel names under which the committed PMC passes (profiles/rNN_pmc.json) list the HBM traffic per launch.
PMC_DEFERRED = "deferred_walk_kernel<0, 32u, false>"  # the headline launch: per-pixel lists, 32 x 32 tiles, rgba8
PMC_TILED = ("light_bins_kernel", "tile_lists_kernel<0>", "deferred_lists_kernel<0>")
PMC_CULL = {"atomic": ("cull_init_cmds", "cull_atomic_kernel"), "deterministic": ("cull_onepass_kernel",)}


def deferred_algorithmic(width, height, lights):
    """Algorithmic work of one resolve: 24 B/pixel (20 read + 4 written, 8-bit output) and
    150 + 117*L FP32 operations per pixel (FMA counted as two)."""
    px = width * height
    return px * 24.0, px * (150.0 + 117.0 * lights)


def cull_algorithmic(n, visible, meshes):
    return n * 32.0 + visible * 4.0 + meshes * 20.0


def profiled_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc.json, newest
    round): counters cannot be collected inside a timed run, so the figures come from the separate
    --pmc runs of this same command (tools/profile_round.sh).  {} when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return {}, None
    try:
        data = json.load(open(files[-1]))
    except (OSError, ValueError):
        return {}, None
    out = {k: v["hbm_traffic_bytes_per_launch"] for k, v in data.items() if "hbm_traffic_bytes_per_launch" in v}
    return out, os.path.relpath(files[-1], ROOT)


def timed(ctx, fn, iters, warmup):
    """Average milliseconds of fn() over `iters` back-to-back calls on the library's stream (HIP events)."""
    for _ in range(warmup):
        fn()
    ctx.wait_idle()
    e0 = ctx.marker()
    for _ in range(iters):
        fn()
    e1 = ctx.marker()
    ctx.wait_idle()
    return e1.elapsed_ms_since(e0) / iters


TILED_WORKLOADS = {  # name -> (width, height, lights, radius): the radius-cutoff lines of the "callers" object
    "tiled_resolve_4k_256": (3840, 2160, 256, 25.0),
    "tiled_resolve_4k_4096": (3840, 2160, 4096, 25.0),
    "tiled_resolve_8k": (7680, 4320, 4096, 25.0),
    "tiled_resolve_8k_discontinuous": (7680, 4320, 4096, 25.0),  # depth uncorrelated between neighbouring pixels
}


def tiled_line(ctx, vkr, synth, frame, K, WU, name, pmc):
    """One radius-cutoff resolve with light lists, timed.  L_eff = mean number of lights a pixel's loop walked, measured
    from the lists of the timed frame; frac = max(algorithmic bytes / t / HBM peak, flops(L_eff) / t / VALU peak)."""
    Wt, Ht, Lt, Rt = TILED_WORKLOADS[name]
    caset = synth.deferred_case(64, 36, Lt, config_index=3)
    if name.endswith("_discontinuous"):
        # the parity tests' G-buffer at full size: a tile's depth range spans hundreds of world units
        noisy = synth.deferred_case(Wt, Ht, Lt, config_index=3)
        gbt = vkr.GBuffer(ctx, Wt, Ht).upload(noisy["gbuf"])
        caset = noisy
        scene = "G-buffer with depth uncorrelated between neighbouring pixels"
    else:
        gbt = vkr.GBuffer(ctx, Wt, Ht)
        camt = synth.reference_camera(Wt, Ht)
        sbuf = ctx.buffer_from(synth.make_spheres(64))
        frame.gbuffer_fill(ctx, gbt, sbuf, frame.make_gbuffer_scene(camt["vp"], camt["inv_vp"], camt["pos"], 64))
        ctx.wait_idle()
        scene = "analytic G-buffer"
    pusht = caset["push"].copy()
    pusht[3] = Rt
    lightst = ctx.buffer_from(caset["lights"])
    outt = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, Wt, Ht)
    targs = vkr.make_deferred_args(gbt, outt, lightst, Lt, pusht, flags=vkr.DEFERRED_RADIUS_CUTOFF)
    ms = timed(ctx, lambda: ctx.deferred_resolve(targs), K, WU)
    st = ctx.tile_stats()
    l_eff = st["mean_lights_per_pixel"]
    byt = Wt * Ht * 24.0
    flops = Wt * Ht * (150.0 + 117.0 * l_eff)
    hbm_frac, valu_frac = byt / (ms * 1e-3) / HBM_PEAK, flops / (ms * 1e-3) / VALU_FP32_PEAK
    line = {"value": Wt * Ht / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms, "lights": Lt, "radius": Rt,
            "workload": f"{Wt}x{Ht}, {Lt} lights, radius cutoff {Rt}, coarse bins + per-tile light lists, {scene}",
            "L_eff": l_eff, "lists": {k: st[k] for k in ("tiles", "dark_tiles", "listed_tiles", "deep_tiles", "overflow_tiles", "max_list", "list_cap")},
            "roofline": {"bound": "hbm" if hbm_frac >= valu_frac else "valu_fp32", "frac": max(hbm_frac, valu_frac), "hbm_frac": hbm_frac,
                         "valu_frac": valu_frac, "algorithmic_bytes": byt, "algorithmic_flops": flops, "traffic": None, "dram_frac": None},
            "hbm_frac": hbm_frac, "algorithmic_bytes": byt}
    # HBM bytes the launches of one call really moved (PMC passes of this workload alone): dark tiles cost 12 of the 24
    # bytes per pixel, so the DRAM rate is below what the algorithmic bytes suggest - both fractions are in the line
    traffic = pmc.get(name)
    if traffic is not None:
        line["roofline"]["traffic"] = traffic
        line["roofline"]["dram_frac"] = traffic / (ms * 1e-3) / HBM_PEAK
    return line, (gbt, outt, lightst, pusht)


def configs0_line(ctx, vkr, synth, K, WU, cpu_cores):
    """The configuration the reference itself runs: a 1920 x 1080 resolve with 32 point lights and a cull of 10 000
    instances, each timed on the GPU and - cpu_cores > 0 - as the CPU restatement on the whole input (it is small),
    with every thread of this job's share and with one."""
    W0, H0, L0, N0, M0 = 1920, 1080, 32, 10_000, 64
    case = synth.deferred_case(W0, H0, L0, config_index=0)
    gb = vkr.GBuffer(ctx, W0, H0).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W0, H0)
    dargs = vkr.make_deferred_args(gb, out, lights, L0, case["push"])
    ms = timed(ctx, lambda: ctx.deferred_resolve(dargs), K, WU)
    by, fl = deferred_algorithmic(W0, H0, L0)
    res = {"workload": f"{W0}x{H0} resolve with {L0} point lights (rgba8) + frustum cull of {N0} instances in {M0} meshes",
           "resolve": {"value": W0 * H0 / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms,
                       "roofline": {"bound": "valu_fp32", "frac": fl / (ms * 1e-3) / VALU_FP32_PEAK, "hbm_frac": by / (ms * 1e-3) / HBM_PEAK,
                                    "algorithmic_flops": fl, "algorithmic_bytes": by}}}
    scene = synth.cull_case(N0, M0, config_index=0)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    res["cull"] = {}
    for mode_name, flags in (("atomic", 0), ("deterministic", vkr.CULL_DETERMINISTIC)):
        cargs = cs.args(scene["planes"], flags)
        cms = timed(ctx, lambda: ctx.prep_indirect(cargs), K, WU)
        vis = int(cs.results()["cmds"]["instanceCount"].sum())
        cb = cull_algorithmic(N0, vis, M0)
        res["cull"][mode_name] = {"value": N0 / (cms * 1e-3) / 1e9, "unit": "G instances/s", "ms_per_call": cms, "visible_fraction": vis / N0,
                                  "roofline": {"bound": "hbm", "frac": cb / (cms * 1e-3) / HBM_PEAK, "algorithmic_bytes": cb,
                                               "note": "a launch of ten workgroups: the call is launch latency, not bandwidth"}}
    if cpu_cores > 0:
        import oracle  # the CPU baseline: never on the measured GPU path
        base = {}
        for name, threads in (("all_cores", cpu_cores), ("one_core", 1)):
            t0 = time.perf_counter()
            o = oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", threads=threads, want=("rgba8",))
            t = time.perf_counter() - t0
            base[name] = {"value": W0 * H0 / t / 1e9, "unit": "Gpixel/s", "cores": threads, "seconds": t}
        got = out.download(np.uint8, 4)
        worst = int(np.abs(got[..., :3].astype(np.int32) - o["rgba8"][..., :3].astype(np.int32)).max())
        base["kind"], base["sample"] = "port", "the whole frame, binary32 C restatement of deferred.comp"
        base["max_8bit_difference_from_the_gpu_frame"] = worst
        res["resolve"]["cpu_baseline"] = base
        reps = 200  # one cull of 10 000 instances takes the scalar code ~60 us: repeated for a measurable time
        t0 = time.perf_counter()
        for _ in range(reps):
            ref = oracle.cull(scene["aabb_min"], scene["aabb_max"], scene["planes"], scene["mesh_base"], scene["cmd_template"], want_mask=False)
        t = (time.perf_counter() - t0) / reps
        res["cull"]["cpu_baseline"] = {"value": N0 / t / 1e9, "unit": "G instances/s", "cores": 1, "kind": "port", "seconds_per_call": t,
                                       "sample": f"all {N0} instances, scalar C restatement, mean of {reps} calls (includes the Python call)",
                                       "commands_equal_the_gpu's": bool(ref["cmds"].tobytes() == cs.results()["cmds"].tobytes())}
    return res


def bench_callers(ctx, vkr, synth, vdist, args, K, WU, rank, world, barrier, max_over_ranks, band_bench, pmc):
    """The steps before the two passes, each on the headline sizes: per-instance bounds (2^22 instances),
    G-buffer attachment encode and analytic fill (4K), and the 64-frame camera path (cull + fill + resolve
    per frame, three frames in flight).  Per-rank figures, every rank running the same work - except the camera path,
    whose 64 frames are dealt out over the ranks (frame f on rank f % world)."""
    import ctypes as C
    frame = importlib.import_module("vk-renderer_amd.frame")
    W, H, L = args.width, args.height, args.lights
    N, M = args.instances, args.meshes
    out = {}

    soup = synth.make_instance_soup(N, M)
    mats, base, mesh = ctx.buffer_from(soup["mesh_matrices"]), ctx.buffer_from(soup["mesh_base"]), ctx.buffer_from(soup["mesh_aabb"])
    mn, mx = ctx.buffer(N * 16), ctx.buffer(N * 16)
    ms = timed(ctx, lambda: vkr._check(vkr.lib().vkr_instance_aabbs(ctx.handle, mats.ptr, base.ptr, mesh.ptr, M, N, mn.ptr, mx.ptr)), K, WU)
    byt = N * 96.0  # 64-byte matrix in, two float4 out
    out["instance_aabbs"] = {"value": N / (ms * 1e-3) / 1e9, "unit": "G instances/s", "ms_per_call": ms,
                             "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                          "frac": byt / (ms * 1e-3) / HBM_PEAK, "traffic": pmc.get("instance_aabb_kernel"), "algorithmic_bytes": byt}}
    del mats, mn, mx

    gb = vkr.GBuffer(ctx, W, H)
    rng = np.random.default_rng(1)
    planes = [ctx.buffer_from(rng.random((H, W, 4), dtype=np.float32)) for _ in range(3)] + [ctx.buffer_from(rng.random((H, W), dtype=np.float32))]
    ms = timed(ctx, lambda: frame.gbuffer_encode(ctx, gb, *planes), K, WU)
    byt = W * H * 72.0  # 52 bytes of float planes in, 20 bytes of attachments out
    out["gbuffer_encode"] = {"value": W * H / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms,
                             "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                          "frac": byt / (ms * 1e-3) / HBM_PEAK, "traffic": pmc.get("gbuffer_encode_kernel"), "algorithmic_bytes": byt}}
    # both shader stages of the G-buffer pass per pixel (tangent frame from the pixel's model matrix + normal map):
    # five float4 planes and depth in, the pixel's matrix from a 4096-entry table (index constant over 64-pixel runs)
    idx = np.repeat(rng.integers(0, 4096, (H, (W + 63) // 64), dtype=np.uint32), 64, axis=1)[:, :W]
    nobj = rng.standard_normal((H, W, 4), dtype=np.float32)
    nobj[..., 3] = idx.view(np.float32)
    planes = [ctx.buffer_from(rng.random((H, W, 4), dtype=np.float32)) for _ in range(3)]
    planes += [ctx.buffer_from(nobj), ctx.buffer_from(rng.standard_normal((H, W, 4), dtype=np.float32)), ctx.buffer_from(rng.random((H, W), dtype=np.float32)),
               ctx.buffer_from(rng.standard_normal((4096, 16), dtype=np.float32))]
    del nobj, idx
    ms = timed(ctx, lambda: frame.gbuffer_encode_surfaces(ctx, gb, *planes, 4096), K, WU)
    byt = W * H * 104.0  # 84 bytes of float planes in, 20 bytes of attachments out (the matrices come from the caches)
    out["gbuffer_encode_surfaces"] = {"value": W * H / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms,
                                      "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                                   "frac": byt / (ms * 1e-3) / HBM_PEAK, "traffic": pmc.get("gbuffer_surface_kernel"), "algorithmic_bytes": byt}}
    del planes

    cam = synth.reference_camera(W, H)
    spheres = synth.make_spheres(64)
    sbuf = ctx.buffer_from(spheres)
    sc = frame.make_gbuffer_scene(cam["vp"], cam["inv_vp"], cam["pos"], 64)
    ms = timed(ctx, lambda: frame.gbuffer_fill(ctx, gb, sbuf, sc), K, WU)
    byt = W * H * 20.0
    # A write stream of 20 bytes per pixel by its algorithmic bytes; what the kernel spends its time on is the ray of
    # each pixel, the hit's attributes and their exact divisions and square roots (the spheres a tile's rays cannot
    # hit are left out per tile, so the sphere tests no longer count: 64 spheres x 9 operations per pixel would be
    # flops_if_every_sphere_were_tested).
    out["gbuffer_fill"] = {"value": W * H / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms, "spheres": 64,
                           "roofline": {"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                        "frac": byt / (ms * 1e-3) / HBM_PEAK, "traffic": pmc.get("gbuffer_fill_kernel"), "algorithmic_bytes": byt,
                                        "flops_if_every_sphere_were_tested": W * H * (110.0 + 9.0 * 64)}}

    # The whole-table resolve (the reference's rule: every light on every pixel) over that analytic G-buffer: the sky's
    # waves have no pixel a light can change and do not walk the table, and over one surface the trips whose lights
    # all lie behind it are cut short (both exact: same bytes).  Reported with the nominal work of the headline formula.
    lights_a = ctx.buffer_from(synth.deferred_case(64, 36, L, config_index=5)["lights"])
    out_a = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    wargs = vkr.make_deferred_args(gb, out_a, lights_a, L, vkr.deferred_push(cam["pos"], cam["inv_vp"]))
    ms = timed(ctx, lambda: ctx.deferred_resolve(wargs), K, WU)
    _, wflops = deferred_algorithmic(W, H, L)
    out["whole_table_analytic"] = {"value": W * H / (ms * 1e-3) / 1e9, "unit": "Gpixel/s", "ms_per_call": ms, "lights": L,
                                   "workload": f"{W}x{H}, {L} lights, no cutoff, analytic G-buffer (sky + ground + 64 spheres)",
                                   "nominal_valu_frac": wflops / (ms * 1e-3) / VALU_FP32_PEAK}
    del lights_a, out_a

    # Radius cutoff with light lists: the 8K x 4096-light configuration and the two 4K ones on the analytic scene
    # (coherent depth), and the 8K one again on depth that is uncorrelated between neighbouring pixels.
    for name in ("tiled_resolve_4k_256", "tiled_resolve_4k_4096", "tiled_resolve_8k_discontinuous"):
        out[name], keep = tiled_line(ctx, vkr, synth, frame, K, WU, name, pmc)
        del keep
    W8, H8, L8, R8 = TILED_WORKLOADS["tiled_resolve_8k"]
    out["tiled_resolve_8k"], (gb8, out8, lights8, push8) = tiled_line(ctx, vkr, synth, frame, K, WU, "tiled_resolve_8k", pmc)
    if world > 1:  # the same 8K frame in screen bands, one band per rank, gathered on rank 0
        out["tiled_resolve_8k"]["band_sharded"] = band_bench(gb8, out8, lights8, L8, push8, vkr.DEFERRED_RADIUS_CUTOFF, W8, H8)
    del gb8, out8, lights8
    # the fill above left the analytic scene in gb; the camera path below fills its own G-buffer

    # camera path: 64 frames, each = cull of the N-instance set + analytic fill + resolve with L lights
    scene = synth.cull_case(N, M, config_index=2 + 10 * rank)
    cs = vkr.CullScene(ctx, scene["aabb_min"], scene["aabb_max"], scene["mesh_base"], scene["cmd_template"])
    lights = synth.deferred_case(64, 36, L, config_index=5)["lights"]
    loop = frame.FrameLoop(ctx, W, H, np.ascontiguousarray(lights, np.float32).reshape(-1, 12), spheres, cs, frames_in_flight=3)
    camera = frame.Camera()
    loop.run_path(camera, 3)  # warm-up
    barrier()
    t0 = time.perf_counter()
    frames = 64
    mine = vdist.frame_slices(frames, world, rank)  # frame f is rendered by rank f % world
    loop.run_path(camera, frames, only=mine)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    out["camera_path"] = {"value": frames / dt, "unit": "frames/s", "ms_per_frame": dt / len(mine) * 1e3, "frames": frames,
                          "frames_per_rank": len(mine), "scaling": "strong",
                          "workload": f"per frame: cull {N} instances, analytic G-buffer fill {W}x{H}, resolve with {L} lights; 3 frames in flight; frame f on rank f % {world}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: after an idle period the part needs about 12 launches of the 4K x 256 resolve (30 ms of load) to
    # settle at its sustained clock - the first launch takes 3.6 ms, the thirteenth 2.4 ms
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=15)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lights", type=int, default=256)
    ap.add_argument("--instances", type=int, default=1 << 22)
    ap.add_argument("--meshes", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cull", action="store_true")
    ap.add_argument("--no-cold", action="store_true", help="skip the run that rotates over four G-buffer sets")
    ap.add_argument("--only-tiled", choices=sorted(TILED_WORKLOADS), help="run this radius-cutoff workload and nothing else (counter passes: the list kernels' per-launch averages then belong to one workload)")
    ap.add_argument("--no-configs0", action="store_true", help="skip the 1080p x 32 + 10k-instance configuration (counter passes: its launches would enter the per-kernel averages of the headline sizes)")
    ap.add_argument("--no-callers", action="store_true", help="skip the steps either side of the passes (bounds, G-buffer producers, camera path)")
    args = ap.parse_args()

    # `python3 bench.py --gpus N` with no rank environment: this process becomes the parent of N fresh ranks (it has
    # not touched the GPU: nothing imported so far does), relays rank 0's line and exits with the ranks' status.
    launch = importlib.import_module("vk-renderer_amd.launch")
    if launch.needs_spawn(args.gpus):
        limit = float(os.environ.get("VKR_BENCH_TIME_LIMIT_S", "1500"))
        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, time_limit_s=limit,
                                            is_result=lambda line: line.startswith('{"metric"')))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # RCCL is the backend.  VKR_BENCH_BACKEND=gloo is a rehearsal of the N>1 control flow on a box with fewer
    # GPUs than ranks: ranks share devices and the collectives go through host memory, so its numbers mean nothing.
    backend = os.environ.get("VKR_BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit(f"VKR_BENCH_BACKEND={backend}: expected nccl or gloo")
    device_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        # a rank that dies or hangs must end the run, not stall it: collectives time out (RCCL's watchdog aborts the
        # communicator and raises in every surviving rank), and main() turns any exception into a non-zero exit
        from datetime import timedelta
        limit = timedelta(seconds=float(os.environ.get("VKR_BENCH_COLLECTIVE_TIMEOUT_S", "120")))
        # 2 = on a collective's timeout the watchdog aborts the communicator and leaves the process alive: whatever waits
        # for the collective returns, the next collective raises, and the sections below record that instead of dying
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "2")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index), timeout=limit)
        else:
            dist.init_process_group("gloo", timeout=limit)

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(seconds):
        if world == 1:
            return seconds
        tt = torch.tensor([seconds], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    vkr = importlib.import_module("vk-renderer_amd")
    synth = importlib.import_module("vk-renderer_amd.synth")
    vdist = importlib.import_module("vk-renderer_amd.dist")
    ctx = vkr.Context(device_index)
    info = ctx.device_info()

    W, H, L = args.width, args.height, args.lights
    K, WU = args.steps, args.warmup
    if args.only_tiled:
        line, _ = tiled_line(ctx, vkr, synth, importlib.import_module("vk-renderer_amd.frame"), K, WU, args.only_tiled, {})
        if rank == 0:
            print(json.dumps({args.only_tiled: line}))
        return

    # ---------------------------------------------------------------- deferred: frame per rank
    case = synth.deferred_case(W, H, L, config_index=1 + rank)
    gb = vkr.GBuffer(ctx, W, H).upload(case["gbuf"])
    lights = ctx.buffer_from(case["lights"])
    out = ctx.image(vkr.FORMAT_R8G8B8A8_UNORM, W, H)
    dargs = vkr.make_deferred_args(gb, out, lights, L, case["push"])

    # ---------------------------------------------------------------- deferred: inputs not in the Infinity Cache
    # One 4K G-buffer set (166 MB) fits in the 256 MB Infinity Cache, so back-to-back launches over one set
    # read it from there.  Rotating over four sets at distinct addresses (664 MB) makes every launch read HBM.
    # This run comes first: after an idle period the part takes about a dozen launches of this kernel to reach its
    # sustained clock, and the headline below (W warm-up steps, then K timed ones) should not be timing that ramp.
    cold_ms = None
    if not args.no_cold:
        sets = [gb] + [vkr.GBuffer(ctx, W, H).upload(case["gbuf"]) for _ in range(3)]
        cargs = [vkr.make_deferred_args(g, out, lights, L, case["push"]) for g in sets]
        for i in range(max(WU, len(sets))):
            ctx.deferred_resolve(cargs[i % len(sets)])
        ctx.wait_idle()
        m0 = ctx.marker()
        for i in range(K):
            ctx.deferred_resolve(cargs[i % len(sets)])
        m1 = ctx.marker()
        ctx.wait_idle()
        cold_ms = m1.elapsed_ms_since(m0) / K
        del sets[1:], cargs

    for _ in range(WU):
        ctx.deferred_resolve(dargs)
    ctx.wait_idle()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    marks = [ctx.marker()]
    for _ in range(K):
        ctx.deferred_resolve(dargs)
        marks.append(ctx.marker())  # an event per step on the library's stream: per-launch durations for median / min
    ctx.wait_idle()
    torch.cuda.synchronize()
    barrier()
    t1 = time.perf_counter()
    wall = t1 - t0
    step_ms = [marks[i + 1].elapsed_ms_since(marks[i]) for i in range(K)]
    kern_ms = marks[-1].elapsed_ms_since(marks[0]) / K
    wall = max_over_ranks(wall)
    pixels_total = float(W) * H * K * world
    value = pixels_total / wall / 1e9
    dbytes, dflops = deferred_algorithmic(W, H, L)
    pmc, pmc_file = profiled_traffic()
    default_workload = (W, H, L, args.instances, args.meshes) == (3840, 2160, 256, 1 << 22, 1024)
    d_traffic = pmc.get(PMC_DEFERRED) if default_workload else None
    ach_flops = dflops / (kern_ms * 1e-3)
    ach_bytes = dbytes / (kern_ms * 1e-3)

    result = RESULT
    result.update({
        "metric": "shaded Gpixel/s at 4K x 256 lights (deferred.comp resolve)",
        "value": value,
        "unit": "Gpixel/s",
        "n_gpus": world,
        "steps": K,
        "warmup": WU,
        "ms_per_step": wall / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"deferred resolve {W}x{H}, {L} point lights, rgba8 output, one frame per GPU per step",
                   "width": W, "height": H, "lights": L, "parallelism": f"frame-batch x{world}", "device": info["name"]},
        "roofline": {"bound": "valu_fp32", "kernel": "deferred_walk_kernel", "achieved": ach_flops / 1e12, "peak": VALU_FP32_PEAK / 1e12,
                     "unit": "TFLOP/s", "frac": ach_flops / VALU_FP32_PEAK, "traffic": d_traffic,
                     "traffic_source": pmc_file if d_traffic is not None else None,
                     "kernel_ms": kern_ms, "kernel_ms_median": float(np.median(step_ms)), "kernel_ms_min": float(min(step_ms)),
                     "algorithmic_flops": dflops, "algorithmic_bytes": dbytes,
                     "hbm_achieved_GBps": ach_bytes / 1e9, "hbm_frac": ach_bytes / HBM_PEAK},
    })
    STATE["headline_done"] = True  # every rank is past the headline's last collective: what follows adds to the line
    if cold_ms is not None:
        result["roofline"]["kernel_ms_rotating_4_sets"] = cold_ms
    # `frac` divides the nominal work - every light on every pixel at the full 117 operations - by the time.  Up to 256
    # lights the kernel evaluates in full only the pairs whose light is not behind the pixel's surface (the others are
    # the ones deferred.comp:43,84 returns 0 for; same bytes): the share of such pairs is counted here on a sample of
    # rows, and `executed_frac` prices what is executed - per pixel 150 + L x 6 (the test) + pairs x 117.
    if 0 < L <= 256:
        rows_s = np.arange(0, H, max(H // 24, 1))
        pos_s = synth.unproject(case["push"][4:20], W, H, case["gbuf"]["depth"])[rows_s]
        n_s = case["gbuf"]["normal"][rows_s][..., :3].astype(np.float64)
        n_s = -n_s / np.linalg.norm(n_s, axis=-1, keepdims=True)
        view_s = case["push"][:3].astype(np.float64) - pos_s
        front_s = (view_s * n_s).sum(-1) > 0.0
        lp_s = np.ascontiguousarray(case["lights"], np.float32).reshape(-1, 12)[:, :3].astype(np.float64)
        ln_s = np.einsum("rwc,lc->rwl", n_s, lp_s) - (pos_s * n_s).sum(-1)[..., None]
        share = float(((ln_s > 0.0) & front_s[..., None]).mean())
        executed = float(W) * H * (150.0 + 6.0 * L + 117.0 * L * share)
        result["roofline"].update({
            "pairs_evaluated_in_full": share, "executed_flops": executed, "executed_frac": executed / (kern_ms * 1e-3) / VALU_FP32_PEAK,
            "frac_note": "frac counts every light on every pixel at 117 operations (the algorithmic count, 150 + 117 L per pixel); the kernel skips, exactly, the pairs "
                         "whose light lies behind the pixel's surface (deferred.comp:43,84), so frac may exceed 1; executed_frac prices what is executed"})

    # ---------------------------------------------------------------- deferred: one frame in bands
    # The pipelines are the package's (vk-renderer_amd/dist.py): double-buffered band / command buffers, the gather
    # of step k behind the compute of step k+1, a buffer rewritten only after the gather that read it was waited for.
    if backend == "nccl" and world > 1:
        transport = vdist.RcclTransport(ctx.stream(), device_index)
    else:  # one rank, or the gloo rehearsal: the kernels fill device tensors, the gather goes through host memory
        transport = vdist.HostTransport(device="cuda", sync=ctx.wait_idle)

    def device_sync():
        ctx.wait_idle()
        torch.cuda.synchronize()

    def band_bench(gbuf, out_img, lights_buf, n_lights, push, flags, width, height):
        """This rank resolves its band of rows of ONE frame and the bands are gathered on rank 0, K times
        (dist.BandShardedResolve).  Afterwards rank 0 checks the gathered frame against its own resolve of the
        whole frame (outside the timed region)."""
        pipe = vdist.BandShardedResolve(transport, width, height, out_img.pitch,
                                        vdist.hip_band_resolver(vkr, ctx, gbuf, out_img, lights_buf, n_lights, push, flags))
        tb = vdist.timed_steps(transport, pipe, K, WU, device_sync)
        got = pipe.frame()
        verified = None
        if rank == 0:
            ctx.deferred_resolve(vkr.make_deferred_args(gbuf, out_img, lights_buf, n_lights, push, flags=flags))
            ctx.wait_idle()
            verified = bool(np.array_equal(got, out_img.download(np.uint8, 4)))
        return {"value": float(width) * height * K / tb / 1e9, "unit": "Gpixel/s", "ms_per_frame": tb / K * 1e3, "scaling": "strong",
                "collective": "gather of rgba8 bands to rank 0 (RCCL), double-buffered: the gather of frame k overlaps the resolve of frame k+1",
                "band_rows": [list(vdist.band_rows(height, world, r)) for r in range(world)], "verified": verified}

    def sharded(name, fn):
        """A sharded pipeline's line, or - when it raises (a collective that fails or times out, a HIP error) - the
        error in its place and in the line's "errors"; once one has failed the others are not started (the
        communicator may be gone), while the per-rank measurements go on."""
        if STATE["sharded_failed"]:
            return {"skipped": "an earlier sharded pipeline failed", "verified": None}
        try:
            if os.environ.get("VKR_BENCH_FAIL") == name:  # rehearsal of this path: VKR_BENCH_FAIL=band_sharded
                raise RuntimeError("failure injected through VKR_BENCH_FAIL")
            return fn()
        except Exception as e:  # noqa: BLE001 - recorded, and the run goes on
            import traceback
            traceback.print_exc()
            STATE["sharded_failed"] = True
            result.setdefault("errors", {})[name] = f"{type(e).__name__}: {e}"
            return {"error": f"{type(e).__name__}: {e}", "verified": None}

    if world > 1:
        frame = synth.deferred_case(W, H, L, config_index=1)  # the same frame on every rank
        gb.upload(frame["gbuf"])
        lights.upload(frame["lights"])
        result["band_sharded"] = sharded("band_sharded", lambda: band_bench(gb, out, lights, L, frame["push"], 0, W, H))

    # ---------------------------------------------------------------- cull
    if os.environ.get("VKR_BENCH_FAIL") == "fatal":  # rehearsal of the exit handler: the line goes out with the headline
        raise RuntimeError("failure injected through VKR_BENCH_FAIL")
    if not args.no_cull:
        N, M = args.instances, args.meshes
        # rotate through distinct input sets larger than the 256 MiB Infinity Cache in total
        nsets = 3
        scenes, css = [], []
        for s in range(nsets):
            sc = synth.cull_case(N, M, config_index=2 + 10 * rank + s)
            scenes.append(sc)
            css.append(vkr.CullScene(ctx, sc["aabb_min"], sc["aabb_max"], sc["mesh_base"], sc["cmd_template"]))
        planes = scenes[0]["planes"]
        cull = {}
        for mode_name, flags in (("atomic", 0), ("deterministic", vkr.CULL_DETERMINISTIC), ("atomic_commands_reset_beforehand", vkr.CULL_CMDS_PREZEROED)):
            cargs = [cs.args(planes, flags) for cs in css]
            for i in range(max(WU, nsets)):
                ctx.prep_indirect(cargs[i % nsets])
            ctx.wait_idle()
            if flags & vkr.CULL_CMDS_PREZEROED:
                # the commands of all three sets are reset outside the timed region, as a frame loop resets them behind
                # its draws; K <= 3 calls per set would be the faithful pattern, so the timed loop runs in rounds of
                # three calls with the resets of the next round between the timed regions
                rounds, call_ms_sum = (K + nsets - 1) // nsets, 0.0
                torch.cuda.synchronize()
                barrier()
                c0 = time.perf_counter()
                for _ in range(rounds):
                    for a_ in cargs:
                        ctx.prep_indirect_reset(a_)
                    e0 = ctx.marker()
                    for a_ in cargs:
                        ctx.prep_indirect(a_)
                    e1 = ctx.marker()
                    ctx.wait_idle()
                    call_ms_sum += e1.elapsed_ms_since(e0)
                torch.cuda.synchronize()
                barrier()
                call_ms = call_ms_sum / (rounds * nsets)
                vis = int(css[0].results()["cmds"]["instanceCount"].sum())
                cb = cull_algorithmic(N, vis, M)
                cull[mode_name] = {"ms_per_call": call_ms, "visible_fraction": vis / N, "flag": "VKR_CULL_CMDS_PREZEROED",
                                   "timed": "the cull launches only (HIP events around each round of three calls); the resets run between the rounds",
                                   "roofline": {"bound": "hbm", "achieved": cb / (call_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                                                "frac": cb / (call_ms * 1e-3) / HBM_PEAK, "traffic": None, "algorithmic_bytes": cb}}
                assert vis == int(cull["atomic"]["visible_fraction"] * N + 0.5)
                continue
            torch.cuda.synchronize()
            barrier()
            c0 = time.perf_counter()
            e0 = ctx.marker()
            for i in range(K):
                ctx.prep_indirect(cargs[i % nsets])
            e1 = ctx.marker()
            ctx.wait_idle()
            torch.cuda.synchronize()
            barrier()
            cw = max_over_ranks(time.perf_counter() - c0)
            call_ms = e1.elapsed_ms_since(e0) / K
            vis = int(css[0].results()["cmds"]["instanceCount"].sum())
            cb = cull_algorithmic(N, vis, M)
            kernels = PMC_CULL[mode_name]
            c_traffic = sum(pmc[k] for k in kernels) if default_workload and all(k in pmc for k in kernels) else None
            cull[mode_name] = {"value": float(N) * K * world / cw / 1e9, "unit": "G instances/s", "ms_per_call": call_ms,
                               "visible_fraction": vis / N,
                               "roofline": {"bound": "hbm", "achieved": cb / (call_ms * 1e-3) / 1e9, "peak": HBM_PEAK / 1e9,
                                            "unit": "GB/s", "frac": cb / (call_ms * 1e-3) / HBM_PEAK, "traffic": c_traffic,
                                            "algorithmic_bytes": cb}}
        cull["config"] = {"workload": f"frustum cull + compaction, {N} instance AABBs, {M} meshes, {nsets} rotating input sets",
                          "parallelism": f"replicated scene x{world}"}
        if world > 1:
            # ONE scene split into instance ranges on mesh boundaries, one range per rank; the only exchange is the
            # gather of the per-mesh commands (20 bytes per mesh), the compacted lists stay where they were made
            def range_bench():
                shared = synth.cull_case(N, M, config_index=2)
                rpipe = vdist.RangeShardedCull(transport, shared["mesh_base"], N, vdist.hip_range_culler(vkr, ctx, shared, shared["planes"], 0))
                cuts = rpipe.cuts
                rt = vdist.timed_steps(transport, rpipe, K, WU, device_sync)
                got = rpipe.commands()
                verified = None
                if rank == 0:  # the assembled command buffer against this rank's cull of the whole scene
                    whole = vkr.CullScene(ctx, shared["aabb_min"], shared["aabb_max"], shared["mesh_base"], shared["cmd_template"])
                    whole.run(shared["planes"], 0)
                    ctx.wait_idle()
                    verified = got == whole.results()["cmds"].tobytes()
                    del whole
                del rpipe
                return {"value": float(N) * K / rt / 1e9, "unit": "G instances/s", "ms_per_call": rt / K * 1e3, "scaling": "strong",
                        "collective": "gather of the per-mesh commands to rank 0 (RCCL), double-buffered",
                        "ranges": [[int(c[2]), int(c[3])] for c in cuts], "verified": verified}

            cull["range_sharded"] = sharded("cull.range_sharded", range_bench)
        result["cull"] = cull

    # ---------------------------------------------------------------- callers either side of the passes
    if not args.no_callers and not args.no_cull:
        result["callers"] = bench_callers(ctx, vkr, synth, vdist, args, K, WU, rank, world, barrier, max_over_ranks,
                                          lambda *a: sharded("callers.tiled_resolve_8k.band_sharded", lambda: band_bench(*a)), pmc if default_workload else {})

    # ---------------------------------------------------------------- CPU baseline (rank 0, N=1)
    # Threads of the CPU legs: every CPU this process may run on.  os.cpu_count() reports the whole machine; the
    # scheduler affinity mask and the thread budget the box exports (OMP_NUM_THREADS) say what this job's share is.
    host_cpus = os.cpu_count() or 1
    share = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cpus
    budget = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(share, budget) if budget > 0 else share)
    cpu_legs = world == 1 and not args.no_cpu_baseline
    if cpu_legs:
        import oracle  # checker/baseline only: never on the measured GPU path
        oracle.build()

    # ---------------------------------------------------------------- the reference's own runnable configuration
    if rank == 0 and not args.no_cull and not args.no_configs0:
        result["configs0"] = configs0_line(ctx, vkr, synth, K, WU, cores if cpu_legs else 0)

    if cpu_legs:
        # a bounded sample: about 20 s of CPU work (the scalar restatement shades ~0.18 Mpixel/s per core at 256 lights)
        rows = int(min(H, max(cores, 20.0 * cores * 0.18e6 * (256.0 / max(L, 1)) / W)))
        tc0 = time.perf_counter()
        oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", rows=(0, rows), threads=cores, want=("rgba8",))
        tc = time.perf_counter() - tc0
        result["cpu_baseline"] = {"value": float(W) * rows / tc / 1e9, "unit": "Gpixel/s", "cores": cores, "kind": "port",
                                  "sample": f"rows 0..{rows} of the same {W}x{H} frame, {L} lights, binary32 C restatement of deferred.comp, {cores} threads",
                                  "seconds": tc, "host_cpus": host_cpus, "cpus_in_affinity_mask": share, "thread_budget_env": budget or None}
        rows1 = max(rows // cores, 1)  # one thread on its share of those rows: about as long as the threaded run
        tc0 = time.perf_counter()
        oracle.deferred(case["gbuf"], case["lights"], case["push"], mode="f32", rows=(0, rows1), threads=1, want=("rgba8",))
        tc1 = time.perf_counter() - tc0
        result["cpu_baseline"]["one_core"] = {"value": float(W) * rows1 / tc1 / 1e9, "unit": "Gpixel/s", "sample": f"rows 0..{rows1}, 1 thread", "seconds": tc1}
        if not args.no_cull:
            from concurrent.futures import ThreadPoolExecutor
            sc = scenes[0]
            tc0 = time.perf_counter()
            oracle.cull(sc["aabb_min"], sc["aabb_max"], planes, sc["mesh_base"], sc["cmd_template"], want_mask=False)
            tc = time.perf_counter() - tc0
            result["cull"]["cpu_baseline"] = {"value": float(N) / tc / 1e9, "unit": "G instances/s", "cores": 1, "kind": "port",
                                              "sample": f"all {N} instances, scalar C restatement", "seconds": tc}
            # all cores: mesh ranges (the multi-GPU partition) culled side by side, the C code runs outside the interpreter lock
            cuts = [c for c in vdist.mesh_range_cuts(sc["mesh_base"], N, cores) if c[1] > c[0]]

            def cull_range(c):
                m0, m1, i0, i1 = c
                sub = sc["aabb_min"][i0:i1].copy()
                sub[:, 3] = (sub[:, 3].view(np.uint32) - np.uint32(m0)).view(np.float32)
                return oracle.cull(sub, sc["aabb_max"][i0:i1], planes, sc["mesh_base"][m0:m1] - np.uint32(i0), sc["cmd_template"][m0:m1], want_mask=False)

            tc0 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                parts = list(ex.map(cull_range, cuts))
            tcm = time.perf_counter() - tc0
            assert sum(int(p_["cmds"]["instanceCount"].sum()) for p_ in parts) == int(css[0].results()["cmds"]["instanceCount"].sum())
            result["cull"]["cpu_baseline"]["all_cores"] = {"value": float(N) / tcm / 1e9, "unit": "G instances/s", "cores": cores,
                                                           "sample": f"all {N} instances in {len(cuts)} mesh ranges, one thread each (includes slicing the inputs)", "seconds": tcm}

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


# What main() has measured so far, and how far it got: the exit handler below prints the line with the headline in
# it when a later section (the sharded pipelines, the callers) ends the run.
RESULT: dict = {}
STATE = {"headline_done": False, "sharded_failed": False}


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:  # a failed collective, a HIP error, a dead peer: say so
        import traceback
        traceback.print_exc()
        rank_ = os.environ.get("RANK", "0")
        print(f"bench.py: rank {rank_} failed: {e}", file=sys.stderr, flush=True)
        if STATE["headline_done"] and not isinstance(e, KeyboardInterrupt):
            # the headline was measured on every rank (its last collective is behind them): the line goes out with what
            # there is and with the failure in it, and the run counts as one that produced its metric
            if rank_ == "0":
                RESULT.setdefault("errors", {})["fatal"] = f"{type(e).__name__}: {e}"
                print(json.dumps(RESULT), flush=True)
            os._exit(0)
        os._exit(1)  # not sys.exit: a hung communicator's destructor must not get the chance to block the exit
