#!/bin/bash
# HBM bytes per launch of the other legs (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, --kernel-trace only):
#   gpurun -- 'bash tools/r2_traffic.sh'   ->  gpurun_out/r2t/traffic.md
O=$PWD/gpurun_out/r2t; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -- python3 $R/tools/legs_once.py > $O/$c.txt 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/r2t/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "kmers::" in n and "synth" not in n:
                acc[n[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
kept = None
for l in open("gpurun_out/r2t/FETCH_SIZE.txt"):
    if l.startswith("U31 kept"):
        kept = int(l.split()[-1])
L = 1e9
alg = {"stream_kernel<2, 2, 1, 1": ("C3 CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2}", 8.25 * 1.25e9),
       "stream_kernel<4, 2, 2, 0": ("C4 FwDNAMers{63} + reverse complements", 32.5 * L),
       "stream_kernel<4, 2, 1, 0, false": ("C5 strict SpacedDNAMers{21,3}", 0.5 * L + 8 * (L - 21) // 3),
       "unambiguous_kernel<4, 1, 0": ("UnambiguousDNAMers{31}, p(N)=0.04", 0.5 * L + 16.0 * (kept or 0)),
       "stream_kernel<8, 2, 1, 1": ("C2 from ASCII text", 17.0 * L)}
out = ["| leg | kernel | FETCH_SIZE KiB x 2 (gfx950) | WRITE_SIZE KiB | HBM bytes per launch | algorithmic bytes | ratio |", "|---|---|---|---|---|---|---|"]
for n, d in acc.items():
    for key, (label, ab) in alg.items():
        if key in n:
            f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
            tot = f * 1024 * 2 + w * 1024
            out.append(f"| {label} | `{n[:60]}` | {f * 2:.0f} | {w:.0f} | {tot / 1e9:.3f} GB | {ab / 1e9:.3f} GB | {tot / ab:.3f} |")
open("gpurun_out/r2t/traffic.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
