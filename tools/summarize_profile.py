#!/usr/bin/env python3
"""Turns the rocprofv3 CSVs under gpurun_out/prof into the small summaries committed in profiles/."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
out = {}


def find(pattern):
    return sorted(glob.glob(os.path.join(root, pattern), recursive=True))


# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = [r for r in rows if "kmers" in r.get("Name", "")][:10]
    out["kernel_stats_top"] = rows[:8]
# per-dispatch durations of the stream kernel
for f in find("trace/**/*kernel_trace.csv"):
    by = {}
    for r in csv.DictReader(open(f)):
        if "kmers::" in r.get("Kernel_Name", ""):
            by.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out["dispatch_durations"] = {}
    for name, d in by.items():
        d.sort()
        out["dispatch_durations"][name] = {"n": len(d), "avg_ns": sum(d) / len(d), "median_ns": d[len(d) // 2],
                                           "min_ns": d[0], "max_ns": d[-1]}


def counters(sub):
    vals = {}
    for f in find(f"{sub}/**/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "kmers::stream_kernel" not in r.get("Kernel_Name", ""):
                continue
            vals.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {kn: {k: {"n": len(v), "avg": sum(v) / len(v)} for k, v in d.items()} for kn, d in vals.items()}


for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    c = counters(sub)
    if c:
        out[sub] = c
json.dump(out, open(os.path.join("profiles", f"{tag}_rocprof_summary.json"), "w"), indent=1)
out.pop("kernel_stats_top", None)
print(json.dumps(out, indent=1)[:8000])
