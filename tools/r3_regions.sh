#!/bin/bash
# Region classes of HBM (profiles/r03_alloc.md):   gpurun -- 'bash tools/r3_regions.sh'   ->  gpurun_out/r3f/
#   1. tools/xcd_affinity: pure fills -- XCD x address matrix, two arrays at (X, X + D), one stream split over two places,
#      k arrays, copy / two read streams                                                                (map_fill.txt)
#   2. C4 / C2 with their two output arrays moved through one 210 GiB block (tools/leg.py --shifts)    (c4_map.txt ...)
#   3. counters of C4 in a slow and in a fast placement (TCC write path, one rocprofv3 --pmc pass per counter set)
O=$PWD/gpurun_out/r3f; rm -rf $O; mkdir -p $O; R=$PWD
[ -x tools/xcd_affinity ] || hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/xcd_affinity tools/xcd_affinity.hip
timeout 600 tools/xcd_affinity 4 200 > $O/map_fill.txt 2>&1
G=$((1<<30))
S=""; for x in $(seq 0 8 160); do S="$S,$((x*G)):$((x*G))"; done; S=${S:1}          # both arrays moved together
python3 tools/leg.py --leg c4 --alloc carve:210 --shifts $S > $O/c4_map.txt 2>> $O/err.txt
S=""; for x in $(seq 0 8 176); do S="$S,0:$((x*G))"; done; S=${S:1}                  # a fixed, b moved away
python3 tools/leg.py --leg c4 --alloc carve:210 --shifts $S > $O/c4_dist.txt 2>> $O/err.txt
python3 tools/leg.py --leg c2 --alloc carve:210 --shifts $S > $O/c2_dist.txt 2>> $O/err.txt
cd /tmp && export TMPDIR=/tmp
for place in 0:0 0:$((64*G)); do
  for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_64B_sum" "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE"; do
    n=$(echo "$place-$set" | tr ' :' '__' | cut -c1-60)
    # (timeout: a counter set the hardware cannot collect aborts the child and leaves rocprofv3 waiting forever)
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
head -60 $O/map_fill.txt; cut -c1-80 $O/c4_map.txt $O/c4_dist.txt $O/c2_dist.txt
