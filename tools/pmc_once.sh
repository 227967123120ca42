#!/bin/bash
# One rocprofv3 --pmc pass over a program, one line per dispatch of the kernels whose name matches $KERNEL (default: all):
#   bash tools/pmc_once.sh <out dir> "<counter> <counter> ..." python3 tools/unamb_once.py
# (counters of one pass must fit the block's slots, MI355X_MICROARCH.md: 8 SQ, 4 TCC, 2 GRBM)
set -u
OUT="$1"; SET="$2"; shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/raw" -- "$@" > "$OUT/run.txt" 2>&1
python3 - "$OUT/raw" "${KERNEL:-}" <<'PY'
import csv, glob, sys, collections
d, pat = sys.argv[1], sys.argv[2]
per, dur = collections.OrderedDict(), {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            e = per.setdefault(int(r["Dispatch_Id"]), {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:70])
for did in sorted(per):
    ns, nm = dur.get(did, (0, "?"))
    print(did, nm, f"{ns / 1e6:.4f} ms", " ".join(f"{k}={v:.4g}" for k, v in sorted(per[did].items())))
PY
rm -rf "$OUT/raw"
