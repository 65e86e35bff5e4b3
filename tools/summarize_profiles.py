#!/usr/bin/env python3
"""Condense a tools/profile_round.sh run (gpurun_out/<tag>_*) into profiles/:
   <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <tag>_bench.json, <tag>_summary.json
   (per-kernel average duration + HBM traffic from the FETCH_SIZE / WRITE_SIZE PMC passes).
HBM traffic per launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes): on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); the x2 is applied to the
streaming kernels (k_sweep_stream, k_rate_table, k_thermal*: 16-B-per-lane loads) and NOT to the narrow-gather
kernels, whose calibration is unknown (reported raw, flagged).  FETCH_SIZE counts the L2's memory-side requests:
Infinity-Cache hits are included, so it is fabric traffic, an upper bound of what reaches HBM."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def kname(raw):
    """cetkmc::k_sweep_stream<true, true, 1, false>(...) -> k_sweep_stream (rate-table variant, any row shape);
    <false, ...> -> k_sweep_stream_recompute; other templates: base name"""
    head = raw.split("(")[0].replace("cetkmc::", "").replace("void ", "")
    base = head.split("<")[0]
    if base == "k_sweep_stream" and "<" in head and head.split("<", 1)[1].strip().startswith("false"):
        return "k_sweep_stream_recompute"
    return base


def one(pattern):
    """the NEWEST match: gpurun merges a re-run's files beside an earlier run's (different PIDs in the names)"""
    g = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return g[-1] if g else None


stats = one(f"{tag}_stats/*/*kernel_stats.csv")
summary = {"tag": tag, "kernels": {}}
if stats:
    shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats)):
        name = kname(r["Name"])
        summary["kernels"][name] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                    "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                    "pct": float(r["Percentage"])}
for which in ("fetch", "write"):
    f = one(f"{tag}_pmc_{which}/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = kname(r["Kernel_Name"])
        agg[name].append(float(r["Counter_Value"]))
    for name, v in agg.items():
        summary["kernels"].setdefault(name, {})[f"{which}_size_kb_median"] = statistics.median(v)
for name, k in summary["kernels"].items():
    if "fetch_size_kb_median" in k and "write_size_kb_median" in k:
        wide = name in ("k_sweep_stream", "k_sweep_stream_recompute", "k_rate_table", "k_thermal", "k_thermal_march", "k_thermal_tiles", "k_thermal_tiles16")
        k["hbm_bytes_per_launch"] = (2.0 if wide else 1.0) * k["fetch_size_kb_median"] * 1024 + k["write_size_kb_median"] * 1024
        k["fetch_x2_applied"] = wide
modes = one(f"{tag}_stats_modes/*/*kernel_stats.csv")
if modes:       # second pass: the incremental and Mode B loops (their kernels only)
    shutil.copy(modes, os.path.join(dst, f"{tag}_kernel_stats_modes.csv"))
    summary["kernels_modes"] = {}
    for r in csv.DictReader(open(modes)):
        summary["kernels_modes"][kname(r["Name"])] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                                                      "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                                      "pct": float(r["Percentage"])}
for which in ("fetch", "write"):          # Mode B / incremental kernels: PMC passes of the command that includes them
    f = one(f"{tag}_modes_pmc_{which}/*/*counter_collection.csv")
    if not f or "kernels_modes" not in summary:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[kname(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    for name, v in agg.items():
        if name in summary["kernels_modes"]:
            summary["kernels_modes"][name][f"{which}_size_kb_median"] = statistics.median(v)
for f in (f"{tag}_bench.json", f"{tag}_bench_under_rocprof.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f))
        summary[f.replace(f"{tag}_", "").replace(".json", "")] = json.loads(open(p).read().strip().splitlines()[-1])
json.dump(summary, open(os.path.join(dst, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in summary["kernels"].items() if "avg_us" in v and v.get("pct", 0) > 1}, indent=1))
