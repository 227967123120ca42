#!/usr/bin/env python3
"""The last two kmers_batch calls of a `rocprofv3 --kernel-trace --output-format csv` run of tools/batch_once.py as a timeline:
every kernel (fills and copies included) between the end of one element kernel and the end of the next, with its start relative to
that end and its duration -- what a call costs on the device besides its element kernel.    python3 tools/call_timeline.py <trace dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find the last ragged_kernel and the launches after the previous ragged_kernel
idx = [i for i, r in enumerate(rows) if "ragged_kernel" in r["Kernel_Name"]]
for which in (-1, -2):
    last, prev = idx[which], idx[which - 1]
    t0 = int(rows[prev]["End_Timestamp"])
    print("--- call")
    for r in rows[prev + 1:last + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{r['Kernel_Name'][:50]:50s} start +{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f} us")
    print(f"first start to last end: {(int(rows[last]['End_Timestamp']) - int(rows[prev + 1]['Start_Timestamp'])) / 1e3:.1f} us")
