#!/usr/bin/env python3
"""After `gpurun -- 'bash tools/r2_evidence.sh'`: files gpurun_out/r2ev/ under profiles/ (run from the repo root on the build host)."""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "r2ev")
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{path}: no JSON line")


bench = last_json_line(os.path.join(E, "bench.json"))
json.dump(bench, open(os.path.join(P, f"{TAG}_bench.json"), "w"))
under = last_json_line(os.path.join(E, "bench_under_rocprof.json"))
json.dump(under, open(os.path.join(P, f"{TAG}_bench_under_rocprof.json"), "w"))
shutil.copy(os.path.join(E, "kernel_stats.csv"), os.path.join(P, f"{TAG}_kernel_stats.csv"))
rf = bench["roofline"]
json.dump({"bases": bench["config"]["bases_per_gpu"], "k": bench["config"]["k"], "src_bits": bench["config"]["src_bits"],
           "traffic_bytes_per_launch": rf["traffic"], "source": rf["traffic_source"],
           "algorithmic_bytes_per_launch": int(rf["bytes_per_kmer"] * rf["kmers_per_launch"])},
          open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)

# kernel durations of rocprofv3 next to the bench's own HIP-event times of the SAME (profiled) run
stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(E, "kernel_stats.csv")))}


def avg_ms(sub, nth=0):
    hits = [r for n, r in stats.items() if sub in n]
    return float(hits[nth]["AverageNs"]) / 1e6, int(hits[nth]["Calls"]), float(hits[nth]["MinNs"]) / 1e6


oc = under["other_configs"]
rows = []
L1 = 999_999_970
# the headline kernel also runs the 10 Gbase leg: split its calls by duration from the kernel trace of the same run
import glob
traces = sorted(glob.glob(os.path.join(E, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(traces[-1]))
        if "stream_kernel<4, 2, 1, 1, true, false, false" in r["Kernel_Name"]]
c2 = [d for d in durs if d < 8.0]
n1 = [d for d in durs if d >= 8.0]
hk_ms = under["roofline"]["kernel_ms"]
rows.append(f"| C2 headline: CanonicalDNAMers{{31}} + fx_hash, 1 Gbase LongDNA{{4}} (2 warm-up + 10 timed launches) | `stream_kernel<4, 2, 1, 1, true, …>` | {len(c2)} | "
            f"{sum(c2) / len(c2):.4f} | {min(c2):.4f} | {hk_ms} | {16.5 * L1 / (sum(c2) / len(c2)) / 1e6 / 8000:.3f} |")
if n1:
    n1_ms = [v.get("kernel_ms") for k, v in oc.items() if k.startswith("N1")]
    rows.append(f"| N1 north star: the same kernel over 10 Gbase LongDNA{{4}} (warm phase + 7 timed launches) | same | {len(n1)} | {sum(n1) / len(n1):.4f} | {min(n1):.4f} | "
                f"{n1_ms[0] if n1_ms else ''} | {16.5 * 9_999_999_970 / (sum(n1) / len(n1)) / 1e6 / 8000:.3f} |")
for label, sub, key, alg in (
        ("C3 shard: CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2}", "stream_kernel<2, 2, 1, 1, true, false, false", "C3", 8.25 * (1_250_000_000 - 30)),
        ("C4: FwDNAMers{63} + reverse_complement, 1 Gbase LongDNA{4}", "stream_kernel<4, 2, 2, 0, true, false, false", "C4", 32.5 * (1_000_000_000 - 62)),
        ("C5 strict: SpacedDNAMers{21,3}, 1 Gbase LongDNA{4}", "stream_kernel<4, 2, 1, 0, false, false, true", "C5 SpacedDNAMers", 0.5e9 + 8.0 * ((1_000_000_000 - 21) // 3 + 1)),
        ("UnambiguousKmers, one pass (both legs: C5 lattice and K = 31)", "unambiguous_kernel<4, 1, 0>", None, None)):
    try:
        ms, calls, mn = avg_ms(sub)
    except IndexError:
        continue
    bench_ms = None
    if key:
        for k, v in oc.items():
            if k.startswith(key):
                bench_ms = v.get("kernel_ms")
    frac = f"{alg / ms / 1e6 / 8000:.3f}" if alg and "headline" not in label else ""
    rows.append(f"| {label} | `{sub}` | {calls} | {ms:.4f} | {mn:.4f} | {bench_ms if bench_ms is not None else ''} | {frac} |")
with open(os.path.join(P, f"{TAG}_configs_rocprof.md"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2` (one run; `{TAG}_kernel_stats.csv`)\n\n"
            "Average and minimum kernel duration as rocprofv3 reports them, next to the HIP-event time `bench.py` printed for the same\n"
            "leg IN THE SAME PROFILED RUN (`" + TAG + "_bench_under_rocprof.json`), and the fraction of 8 TB/s the rocprofv3 average gives on\n"
            "the algorithmic bytes of SURVEY.md section 8(d).  The legs of `other_configs` keep the device busy with the same call for 50 ms\n"
            "before their timed launches (the first ~10 ms after an idle gap run slower, profiles/r02_tuning.md section 1): rocprofv3's average\n"
            "includes those warm-phase launches, bench.py's figure is the median of the 7 timed ones.\n\n"
            "| leg | kernel | calls | rocprofv3 avg ms | min ms | bench.py ms (same run) | frac of 8 TB/s (rocprofv3 avg) |\n|---|---|---|---|---|---|---|\n")
    f.write("\n".join(rows) + "\n\n")
    hk = under["roofline"]
    f.write(f"Headline kernel in the profiled run: bench.py HIP events {hk['kernel_ms']} ms per launch (10 timed launches), "
                f"frac {hk['frac']}; the un-profiled driver-style run of the same box: {bench['roofline']['kernel_ms']} ms, frac {bench['roofline']['frac']}, "
                f"traffic {bench['roofline']['traffic']} B ({bench['roofline']['traffic_source']}).\n")
print(open(os.path.join(P, f"{TAG}_configs_rocprof.md")).read())
