#!/bin/bash
O=$PWD/gpurun_out/r3f; rm -rf $O; mkdir -p $O
timeout 300 tools/xcd_affinity 4 200 2>&1 | sed -n '/experiment 4/,/experiment 1/p' > $O/map_fill.txt
cat $O/map_fill.txt
# the counters of a bad and of a good placement of C4's two arrays inside one block (same process layout, fresh process each)
cd /tmp && export TMPDIR=/tmp; R=$OLDPWD
G=$((1<<30))
for place in 0:0 0:$((64*G)); do
  for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
    n=$(echo "$place-$set" | tr ' :' '__' | cut -c1-60)
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/leg.py --leg c4 --alloc carve:200 --once --shifts $place > $O/pmc_$n.txt 2>&1
  done
done
cd $R
python3 - <<'PY'
import csv, glob, collections, os
O = "gpurun_out/r3f"
for d in sorted(glob.glob(O + "/pmc_*/")):
    acc = collections.defaultdict(list); dur = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stream_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stream_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(os.path.basename(d.rstrip("/")), "ms", [round(x, 3) for x in dur], {k: [f"{x:.4g}" for x in v] for k, v in acc.items()})
PY
