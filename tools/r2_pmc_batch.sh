O=$PWD/gpurun_out/r2v; rm -rf $O; mkdir -p $O; R=$PWD
python3 tools/batch_once.py 0,1,2,3,4,6,8 > $O/rate.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/p1 -- python3 $R/tools/batch_once.py > $O/p1.txt 2>&1
cd $R
cat $O/rate.txt
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r2v/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "ragged_kernel" in n or "recode" in n:
            acc[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    m = lambda k: sum(d[k]) / max(1, len(d[k]))
    act = m("GRBM_GUI_ACTIVE") / 8
    print(n, f"VALU {m('SQ_INSTS_VALU'):.3g} -> issue share {m('SQ_INSTS_VALU') * 4 / 1024 / act:.2f}; waves {m('SQ_WAVES'):.3g}; wait_inst/wave_cycles {m('SQ_WAIT_INST_ANY') / m('SQ_WAVE_CYCLES'):.2f}; wait_any {m('SQ_WAIT_ANY') / m('SQ_WAVE_CYCLES'):.2f}; LDS insts {m('SQ_INSTS_LDS'):.3g}; active cycles {act:.3g}")
PY
