#!/bin/bash
# Where the output arrays live vs the write rate of C2 / C4, with the address-translation and write-path counters of both
# states.  Fresh box:   gpurun -- 'bash tools/r3_alloc.sh'   ->  gpurun_out/r3a/
#   state "slow": plain hipMalloc allocations, first thing on a fresh box
#   state "fast": the same launch with its outputs from the context's arena (one block of 128 GB)
O=$PWD/gpurun_out/r3a; rm -rf $O; mkdir -p $O; R=$PWD
T=$O/times.txt
for leg in c2 c4; do python3 tools/leg.py --leg $leg --alloc plain >> $T 2>> $O/err.txt; done
cd /tmp && export TMPDIR=/tmp
pmc() {  # name, counters...
  local name=$1; shift
  # (timeout: a counter set the hardware cannot collect aborts the child and leaves rocprofv3 waiting forever)
  timeout 180 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -- python3 $R/tools/leg.py --leg $LEG --alloc $ALLOC --once > $O/$name.txt 2>&1
}
passes() {  # state label
  pmc $1_${LEG}_utcl1 TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
  pmc $1_${LEG}_utcl1s TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_LFIFO_FULL_sum
  pmc $1_${LEG}_tccw TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
  pmc $1_${LEG}_lat TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
  pmc $1_${LEG}_grbm GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE
  pmc $1_${LEG}_tcc2 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_BUSY_sum TCC_TAG_STALL_sum
}
ALLOC=plain
for LEG in c2 c4; do passes slow; done
cd $R
for leg in c2 c4; do python3 tools/leg.py --leg $leg --alloc plain >> $T 2>> $O/err.txt; done   # still slow after the passes?
for leg in c2 c4; do python3 tools/leg.py --leg $leg --alloc carve:128 >> $T 2>> $O/err.txt; done
for leg in c2 c4; do python3 tools/leg.py --leg $leg --alloc arena:128 >> $T 2>> $O/err.txt; done
cd /tmp
ALLOC=arena:128
for LEG in c2 c4; do passes fast; done
cd $R
for leg in c2 c4 c5 u31; do python3 tools/leg.py --leg $leg --alloc plain >> $T 2>> $O/err.txt; done  # the sticky state
for leg in c2 c4 c4t c3 c5 u31 u21; do python3 tools/leg.py --leg $leg --alloc arena:0 >> $T 2>> $O/err.txt; done
cat $T
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r3a"
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(O + "/*/")):
    name = os.path.basename(d.rstrip("/"))
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stream_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    state, leg, _ = name.split("_", 2)
    for k, v in acc.items():
        rows[(leg, k)][state] = sum(v) / len(v)
out = ["| leg | counter | slow (plain hipMalloc, fresh box) | fast (arena, 128 GB block) | fast / slow |", "|---|---|---|---|---|"]
for (leg, k), d in sorted(rows.items()):
    s, f = d.get("slow"), d.get("fast")
    out.append(f"| {leg} | {k} | {s:.4g} | {f:.4g} | {f / s if s else float('nan'):.3f} |" if s is not None and f is not None else f"| {leg} | {k} | {s} | {f} | |")
open(O + "/counters.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
tail -5 $O/err.txt
