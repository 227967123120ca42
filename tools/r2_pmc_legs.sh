#!/bin/bash
# VALU issue share of every materialising leg (is any of them bound by instructions rather than by HBM?):  gpurun -- 'bash tools/r2_pmc_legs.sh'
O=$PWD/gpurun_out/r2w; rm -rf $O; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/legs_once.py > $O/p1.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r2w/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kmers::" in n and "synth" not in n:
            acc[n[:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("| kernel | VALU wave-instructions | VALU issue share of the active cycles | waiting share of the wave cycles |")
print("|---|---|---|---|")
for n, d in acc.items():
    m = lambda k: sum(d[k]) / max(1, len(d[k]))
    act = m("GRBM_GUI_ACTIVE") / 8
    print(f"| `{n}` | {m('SQ_INSTS_VALU'):.3g} | {m('SQ_INSTS_VALU') * 4 / 1024 / act:.2f} | {m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'):.2f} |")
PY
