#!/bin/bash
# VALU / LDS instruction counts of the fused consumers against the issue rate:  gpurun -- 'bash tools/r2_fused_pmc.sh'
O=$PWD/gpurun_out/r2u; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/fused_once.py > $O/p1.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r2u/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kmers::" in n and "synth" not in n:
            acc[n[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/r2u/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kmers::" in n and "synth" not in n:
            dur[n[:70]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print("| kernel | ms (under the profiler) | VALU wave-instructions | per kmer-lane | VALU issue time = x 4 cycles / 1024 SIMDs | share of the XCD-active cycles | LDS instructions |")
print("|---|---|---|---|---|---|---|")
for n, d in acc.items():
    m = lambda k: sum(d[k]) / max(1, len(d[k]))
    valu, act = m("SQ_INSTS_VALU"), m("GRBM_GUI_ACTIVE") / 8
    cyc = valu * 4 / 1024
    print(f"| `{n}` | {sum(dur[n]) / max(1, len(dur[n])):.3f} | {valu:.3g} | {valu * 64 / 1e9:.1f} | {cyc:.3g} cycles | {cyc / act:.2f} | {m('SQ_INSTS_LDS'):.3g} |")
PY
