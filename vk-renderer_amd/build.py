"""Builds libvkr_hip.so (gfx950 only) in-tree with hipcc.  No GPU is needed to build."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libvkr_hip.so")

# Both kernel files are built with contraction off: the cull plane test and the deferred
# position reconstruction must match a binary32 evaluation without fused multiply-adds; the
# deferred light loop asks for its FMAs explicitly.
COMMON = ["--offload-arch=gfx950", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(ROOT, "include"), "-I", CSRC]
UNITS = [
    # -fno-slp-vectorize: v_pk_*_f32 issues at half the rate of the scalar forms on gfx950, so packing
    # two pixels into one instruction gains nothing and costs ~6 % here.
    # -falign-loops=32: loop headers on 32-byte boundaries (harmless; the placement sensitivity once seen
    # here was the transcendental issue hazard that the kernel now avoids explicitly).
    ("deferred.hip", ["-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-falign-loops=32"]),
    ("cull.hip", ["-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]),
    ("scene.hip", ["-O3", "-ffp-contract=off", "-fno-slp-vectorize"]),
    ("gbuffer.hip", ["-O3", "-ffp-contract=off", "-fno-slp-vectorize"]),
    ("vkr_api.hip", ["-O2"]),
    ("vkr_host.hip", ["-O2", "-ffp-contract=off"]),
]


def _stale(out: str, deps: list[str]) -> bool:
    return not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps)


def _fingerprint(deps: list[str], cmd: list[str]) -> str:
    """Contents of everything an object is made from, and the command line: time stamps alone have let an object
    that was older than its source in content, but not in time, through."""
    import hashlib
    h = hashlib.sha256(" ".join(cmd).encode())
    for d in sorted(deps):
        with open(d, "rb") as f:
            h.update(d.encode() + b"\0" + f.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "vkr_c.h")]
    objs = []
    for src, flags in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIB_DIR, src.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, "-c", s, "-o", o] + COMMON + flags + (extra_flags or [])
        stamp, want = o + ".stamp", _fingerprint([s] + headers, cmd)
        have = open(stamp).read().strip() if os.path.exists(stamp) else ""
        if force or have != want or _stale(o, [s, __file__] + headers):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
            with open(stamp, "w") as f:
                f.write(want + "\n")
    if force or _stale(LIB_PATH, objs):
        cmd = [hipcc, "-shared", "-o", LIB_PATH, "--offload-arch=gfx950"] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
