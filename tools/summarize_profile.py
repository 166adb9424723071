#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (written by tools/profile_round.sh) into profiles/<tag>_*.

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py`
  profiles/<tag>_pmc.json           per-kernel counter averages + HBM traffic per launch
HBM traffic follows the gfx950 rules: FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE counts a wide
coalesced read stream at half its bytes (128-byte requests tallied as 64), so it is doubled; WRITE_SIZE
is exact for 16-byte-per-lane stores.  The two were collected in separate passes.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def newest(pattern):
    """gpurun merges every run's files into the same directory: take the most recent match."""
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
bench_line = [l for l in open(os.path.join(src, "bench_under_trace.log")) if l.startswith("{")]
if bench_line:
    open(os.path.join(dst, f"{tag}_bench_under_trace.json"), "w").write(bench_line[-1])
if os.path.isdir(os.path.join(src, "trace_headline")):  # the headline part alone (bench.py --no-callers)
    shutil.copy(newest(os.path.join(src, "trace_headline", "*", "*kernel_stats.csv")), os.path.join(dst, f"{tag}_kernel_stats_headline.csv"))
    line = [l for l in open(os.path.join(src, "bench_headline_under_trace.log")) if l.startswith("{")]
    if line:
        open(os.path.join(dst, f"{tag}_bench_headline_under_trace.json"), "w").write(line[-1])

def headline_launches():
    """The headline kernel's launches in the trace of `bench.py --no-callers --no-cpu-baseline`, in start order: the
    run over four rotating G-buffer sets (max(W, 4) warm-up launches and K timed ones), then W warm-up launches and
    K timed ones (the bench line's kernel_ms is the HIP-event mean of exactly those).  The stats file's average is over all of them, cold-clock launches
    included; the mean of the K timed launches is the one to hold against the bench line."""
    line_path = os.path.join(dst, f"{tag}_bench_headline_under_trace.json")
    trace = glob.glob(os.path.join(src, "trace_headline", "*", "*kernel_trace.csv"))
    if not trace or not os.path.exists(line_path):
        return
    line = json.load(open(line_path))
    rows = [r for r in csv.DictReader(open(max(trace, key=os.path.getmtime))) if "deferred_walk_kernel<0, 32u, false>" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    w, k = line["warmup"], line["steps"]
    first = (max(w, 4) + k) if "kernel_ms_rotating_4_sets" in line["roofline"] else 0  # the rotating-sets run comes first
    timed = dur[first + w:first + w + k]
    res = {"kernel": "deferred_walk_kernel<0, 32u, false>", "launches_in_trace": len(dur), "mean_all_launches_ms": sum(dur) / len(dur) / 1e6,
           "timed_steps": {"launches": len(timed), "mean_ms": sum(timed) / len(timed) / 1e6, "median_ms": sorted(timed)[len(timed) // 2] / 1e6, "min_ms": min(timed) / 1e6},
           "first_launches_ms": [d / 1e6 for d in dur[:max(w, 4)]],
           "bench_line_kernel_ms": line["roofline"]["kernel_ms"], "bench_line_kernel_ms_median": line["roofline"]["kernel_ms_median"]}
    res["timed_mean_over_bench_kernel_ms"] = res["timed_steps"]["mean_ms"] / res["bench_line_kernel_ms"]
    json.dump(res, open(os.path.join(dst, f"{tag}_headline_kernel.json"), "w"), indent=1)
    print("headline kernel:", res["timed_steps"], "bench line:", res["bench_line_kernel_ms"])


headline_launches()

out = collections.defaultdict(dict)
for f in [newest(os.path.join(d, "*", "*counter_collection.csv")) for d in glob.glob(os.path.join(src, "pmc_*")) if os.path.isdir(d)]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if k.startswith("__amd"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, cs in agg.items():
        flat = k.replace("(anonymous namespace)::", "")
        m = re.search(r"(\w+)(<[^>(]*>)?\(", flat)  # function name (+ template arguments) before the parameter list
        short = (m.group(1) + (m.group(2) or "")) if m else k
        for c, v in cs.items():
            # skip the first dispatches (clock ramp / cold caches)
            vv = v[len(v) // 3:]
            out[short][c] = sum(vv) / len(vv)
        d = dur[k][len(dur[k]) // 3:]
        out[short].setdefault("duration_us_under_pmc", {})[os.path.basename(os.path.dirname(os.path.dirname(f)))] = sum(d) / len(d) / 1e3

for k, c in out.items():
    if "FETCH_SIZE" in c:
        c["hbm_read_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c:
        c["hbm_write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
    if "hbm_read_bytes_per_launch" in c and "hbm_write_bytes_per_launch" in c:
        c["hbm_traffic_bytes_per_launch"] = c["hbm_read_bytes_per_launch"] + c["hbm_write_bytes_per_launch"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    if "GRBM_GUI_ACTIVE" in c and "pmc_SQ_WAVES" in c.get("duration_us_under_pmc", {}):
        c["clock_GHz_under_pmc"] = c["GRBM_GUI_ACTIVE"] / 8 / (c["duration_us_under_pmc"]["pmc_SQ_WAVES"] * 1e3)

# the radius-cutoff workloads, each profiled alone (bench.py --only-tiled NAME): bytes of one call = sum over its launches
TILED_KERNELS = ("light_bins_kernel", "tile_lists_kernel", "deferred_lists_kernel")
for d in sorted(glob.glob(os.path.join(src, "wl_*__*"))):
    if not os.path.isdir(d):
        continue
    wl, counter = os.path.basename(d)[3:].split("__")
    per_kernel = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(os.path.join(d, "*", "*counter_collection.csv")))):
        if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in TILED_KERNELS):
            per_kernel[next(k for k in TILED_KERNELS if k in r["Kernel_Name"])].append(float(r["Counter_Value"]))
    scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0  # KiB; FETCH_SIZE doubled (see above)
    w = out[wl]
    for k, v in per_kernel.items():
        vv = v[len(v) // 3:]
        w.setdefault("per_kernel_bytes", {}).setdefault(k, {})["read" if counter == "FETCH_SIZE" else "write"] = sum(vv) / len(vv) * scale
for wl, w in out.items():
    if "per_kernel_bytes" in w:
        w["hbm_traffic_bytes_per_launch"] = sum(sum(k.values()) for k in w["per_kernel_bytes"].values())
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
