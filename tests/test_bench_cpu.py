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
    for k in b.PMC_TILED + b.PMC_CULL["atomic"] + b.PMC_CULL["deterministic"]:
        assert k in pmc, k
    tiled = sum(pmc[k] for k in b.PMC_TILED)  # 8K x 4096 with light lists: dark tiles never read normals and material
    assert 0.5 < tiled / (7680 * 4320 * 24) < 0.7


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "torch.distributed.run" in (r.stderr + r.stdout)
