O=$PWD/gpurun_out/r2g; mkdir -p $O; R=$PWD
python3 tools/c4_rate.py > $O/rate.txt 2>&1
python3 tools/c4_rate.py --k 31 >> $O/rate.txt 2>&1
python3 tools/c4_rate.py --k 33 >> $O/rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/tools/c4_rate.py --cases fwrv --reps 2 > $O/pmc1.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT64 --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/tools/c4_rate.py --cases fwrv --reps 2 > $O/pmc2.txt 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/tools/c4_rate.py --cases fwrv --reps 2 > $O/pmc3.txt 2>&1
cd $R
cat $O/rate.txt
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r2g/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "stream_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, f"{sum(v) / len(v):.4g}", len(v))
PY
