"""CPU checks of bench.py: the algorithmic work figures the roofline fractions are computed from, and the loud
exit without a GPU (the bench has no CPU fallback)."""
import importlib.util
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_work_figures():
    b = _bench()
    by, fl = b.deferred_algorithmic(3840, 2160, 256)
    assert by == 3840 * 2160 * 24 == 199_065_600            # 20 B read + 4 B written per pixel
    assert fl == 3840 * 2160 * (150 + 117 * 256)             # 249.7 GFLOP per 4K x 256 frame
    assert abs(fl / 1e9 - 249.68) < 0.01
    assert b.cull_algorithmic(1 << 22, 600_000, 1024) == (1 << 22) * 32 + 600_000 * 4 + 1024 * 20
    assert b.HBM_PEAK == 8.0e12 and abs(b.VALU_FP32_PEAK - 256 * 4 * 32 * 2 * 2.4e9) < 5e10


def test_profiled_traffic_reads_the_committed_profile():
    b = _bench()
    pmc, name = b.profiled_traffic()
    assert name is not None and name.startswith("profiles/") and name.endswith("_pmc.json")
    d = pmc[b.PMC_DEFERRED]
    assert 0.99 < d / 199_065_600 < 1.05  # measured HBM traffic of the 4K x 256 resolve equals its algorithmic bytes
    # every kernel the bench line takes its "traffic" from is in the committed profile under the name bench.py uses
    for k in b.PMC_CULL["atomic"] + b.PMC_CULL["deterministic"]:
        assert k in pmc, k
    if "tiled_resolve_8k" in pmc:  # profiles from this layout on: one entry per radius-cutoff workload
        for name, (w, h, _, _) in b.TILED_WORKLOADS.items():
            # dark tiles never read normals and material; a frame of deep tiles reads depth twice and writes and reads its
            # pixels' depth order on top of the 24 bytes (measured 1.57)
            assert 0.3 < pmc[name] / (w * h * 24) < (1.7 if name.endswith("_discontinuous") else 1.3), name
    else:
        tiled = sum(pmc[k] for k in b.PMC_TILED)
        assert 0.5 < tiled / (7680 * 4320 * 24) < 0.7


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)
    # --gpus 2 without a rank environment: bench.py starts its own two ranks, both fail loudly, and so does the parent
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout) and "exited with status" in r.stderr
    # under a launcher that set another world size
    env = dict(os.environ, RANK="0", WORLD_SIZE="4")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stderr + r.stdout)


RANK_SCRIPT = r"""
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(rank)
mode = sys.argv[1]
if mode == "ok":
    print("noise from rank", rank) if rank else print("library chatter\n" + json.dumps({"world": world, "pad": "x" * 200000}))
elif mode == "fail" and rank == 1:
    sys.exit(7)
elif mode == "fail":
    time.sleep(600)  # a rank waiting for a dead peer: the parent must end it
elif mode == "hang":
    time.sleep(600)
"""


def test_rank_launcher(tmp_path):
    """The parent of `bench.py --gpus N`: N ranks with the rendezvous environment, rank 0's output relayed (also
    when it is longer than a pipe holds), a failing rank or the time limit ends the run with a non-zero status."""
    import importlib
    import io
    import json
    import time
    launch = importlib.import_module("vk-renderer_amd.launch")
    assert launch.needs_spawn(2, {}) and not launch.needs_spawn(1, {})
    assert not launch.needs_spawn(2, {"RANK": "0", "WORLD_SIZE": "2"}) and not launch.needs_spawn(2, {"WORLD_SIZE": "2"})
    script = tmp_path / "rank.py"
    script.write_text(RANK_SCRIPT)
    out, err = io.StringIO(), open(tmp_path / "err.txt", "w+")
    assert launch.spawn_ranks([str(script), "ok"], 3, time_limit_s=120, out=out, err=err, is_result=lambda l: l.startswith("{")) == 0
    line = json.loads(out.getvalue())  # one line: what is not a result went to err
    assert line["world"] == 3 and len(line["pad"]) == 200000 and "noise" not in out.getvalue()
    t0 = time.monotonic()
    assert launch.spawn_ranks([str(script), "fail"], 2, time_limit_s=120, out=io.StringIO(), err=err) == 7
    assert time.monotonic() - t0 < 60
    t0 = time.monotonic()
    assert launch.spawn_ranks([str(script), "hang"], 2, time_limit_s=2, out=io.StringIO(), err=err) == 124
    assert time.monotonic() - t0 < 60
    err.seek(0)
    text = err.read()
    assert "rank 1 exited with status 7" in text and "time limit" in text and "library chatter" in text
