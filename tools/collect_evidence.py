#!/usr/bin/env python3
"""After `gpurun -- 'bash tools/refresh_evidence.sh'`: turns gpurun_out/evidence + gpurun_out/prof into the files
committed under profiles/ (run from the repo root, on the build host)."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E = os.path.join(ROOT, "gpurun_out", "evidence")
P = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"{path}: no JSON line")


def clean(path):
    """Tool output without the HIP runtime's noise lines."""
    return "".join(l for l in open(path) if "amdgpu.ids" not in l)


# 1. rocprofv3: summary JSON, the kernel-stats CSV as rocprofv3 wrote it, the bench line measured under the profiler
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), os.path.join(ROOT, "gpurun_out", "prof"), TAG],
               check=True, cwd=ROOT, stdout=subprocess.DEVNULL)
stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof", "trace", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(P, f"{TAG}_kernel_stats.csv"))
json.dump(last_json_line(os.path.join(ROOT, "gpurun_out", "prof", "trace_bench.json")), open(os.path.join(P, f"{TAG}_bench_under_rocprof.json"), "w"))

# 2. HBM traffic of the headline kernel from the two PMC passes (MI355X_MICROARCH.md: FETCH_SIZE in KiB and halved on
#    gfx950 for coalesced streaming reads, WRITE_SIZE in KiB)
summ = json.load(open(os.path.join(P, f"{TAG}_rocprof_summary.json")))
kern = next(iter(summ["pmc_fetch"]))
fetch, write = summ["pmc_fetch"][kern]["FETCH_SIZE"]["avg"], summ["pmc_write"][kern]["WRITE_SIZE"]["avg"]
old = json.load(open(os.path.join(P, "pmc_traffic.json")))
n_kmers = old["bases"] - old["k"] + 1
traffic = {**old, "kernel": kern, "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "fetch_bytes_corrected": fetch * 1024 * 2,
           "write_bytes": write * 1024, "traffic_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),
           "algorithmic_bytes_per_launch": int(16.5 * n_kmers)}
json.dump(traffic, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)

# 3. bench line (after pmc_traffic.json: `roofline.traffic` is read from it on the box, so it shows the previous
#    collection; the value committed here carries the new one)
bench = last_json_line(os.path.join(E, "bench.json"))
bench["roofline"]["traffic"] = traffic["traffic_bytes_per_launch"]
json.dump(bench, open(os.path.join(P, f"{TAG}_bench.json"), "w"))

# 4. rates of every other entry point
for src, dst in (("other_rates.txt", f"{TAG}_other_rates_final.txt"), ("fused_rates.txt", f"{TAG}_fused_rates_final.txt"),
                 ("batch_rates.txt", f"{TAG}_batch_rates.txt"), ("sketch_lowcomplexity.txt", f"{TAG}_sketch_lowcomplexity.txt"),
                 ("call_latency.txt", f"{TAG}_call_latency.txt"), ("reference_benchmark_10m.txt", f"{TAG}_reference_benchmark_10m.txt"),
                 ("reference_benchmark_1g.txt", f"{TAG}_reference_benchmark_1g.txt")):
    open(os.path.join(P, dst), "w").write(clean(os.path.join(E, src)))
# (the per-record sketch file keeps its history of versions below the current numbers)
cur = clean(os.path.join(E, "sketch_batch_rates.txt"))
path = os.path.join(P, f"{TAG}_sketch_batch_rates.txt")
hist = open(path).read()
marker = "\nbefore (hashes of every record written to HBM"
open(path, "w").write("kmers_minhash_batch, s = 1000, CanonicalDNAMers{16}, 4-bit pool resident, spans resident (tools/sketch_batch_rate.py)\n\n"
                      "fused (record_sketch_kernel: recode pass + one workgroup per record deriving its hashes tile by tile):\n" + cur +
                      (hist[hist.index(marker):] if marker in hist else ""))
r = bench["roofline"]
print(f"headline {bench['value']} {bench['unit']}  frac {r['frac']}  kernel_ms {r['kernel_ms']}  traffic {r['traffic']}")
st = [l for l in open(os.path.join(P, f'{TAG}_kernel_stats.csv')) if 'stream_kernel<4, 2, 1, 1, true, false>' in l]
print(st[0][:160] if st else "no stream kernel row")
