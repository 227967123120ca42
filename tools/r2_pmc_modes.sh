O=$PWD/gpurun_out/r2h; mkdir -p $O; R=$PWD
python3 tools/unamb_modes.py > $O/modes.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/tools/unamb_modes.py > $O/pmc1.txt 2>&1
cd $R
cat $O/modes.txt
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r2h/pmc1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "unambiguous_kernel" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: f"{sum(v) / len(v):.4g}" for c, v in d.items()}, len(next(iter(d.values()))))
PY
