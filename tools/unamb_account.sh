#!/bin/bash
# Per-phase account of unambiguous_kernel<4,1,EMIT> (profiles/r04_unamb.md): SQ counters of the product build and of the builds
# that end behind the stage (-DKMERS_UCUT=1), behind the whole front (2) and behind the look-back (3), on the C5 lattice and at
# K = 31; then the in-kernel stamps of the -DKMERS_STAMPS build.  Variant libraries are built beforehand (they travel with the tree):
#   for c in 1 2 3; do python -m kmers_jl_amd.build variant ucut$c -DKMERS_UCUT=$c unambiguous_api.hip; done
#   python -m kmers_jl_amd.build variant stamps -DKMERS_STAMPS unambiguous_api.hip
# Run on the GPU box from the repo root: gpurun --timeout 1500 -- 'bash tools/unamb_account.sh <tag>'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${1:-r4acc}"
E="$ROOT/gpurun_out/$TAG"
rm -rf "$E"; mkdir -p "$E"
CS="$ROOT/kmers.jl_amd/csrc"
cd /tmp && export TMPDIR=/tmp
SETA="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
SETB="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE"
rocprofv3 -L > "$E/counters.txt" 2>&1
for v in "" ucut1 ucut2 ucut3 ucut4 ucut5 ucut6; do
  lib="$CS/libkmers_hip.so"; name=full
  if [ -n "$v" ]; then lib="$CS/libkmers_hip_$v.so"; name=$v; fi
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  export KMERS_HIP_LIB="$lib"
  i=0
  for set in "$SETA" "$SETB"; do
    i=$((i + 1))
    [ "$name" != full ] && [ $i = 2 ] && continue
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$E/pmc_${name}_$i" -- python3 "$ROOT/tools/unamb_once.py" > "$E/pmc_${name}_$i.txt" 2>&1
    python3 - "$E/pmc_${name}_$i" "$name set $i" <<'PY'
import csv, glob, sys, collections
d, label = sys.argv[1], sys.argv[2]
per = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "unambiguous_kernel" not in r["Kernel_Name"] or "ELi0EEE" not in r["Kernel_Name"].replace(" ", ""):
            pass
        if "unambiguous_kernel" in r["Kernel_Name"]:
            per.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = per.get(int(r["Dispatch_Id"]), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
dur = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:60])
for did in sorted(per):
    ns, nm = dur.get(did, (0, "?"))
    print(label, did, nm, f"{ns / 1e6:.4f} ms", " ".join(f"{k}={v:.4g}" for k, v in sorted(per[did].items())))
PY
  done
done > "$E/account.txt" 2>&1
unset KMERS_HIP_LIB
cat "$E/account.txt"
cd "$ROOT"
if [ -f "$CS/libkmers_hip_stamps.so" ]; then
  KMERS_STAMPS_LIB="$CS/libkmers_hip_stamps.so" KMERS_STAMPS_TILES=49152 KMERS_STAMPS_CASES=2 python3 tools/unamb_stamps.py > "$E/stamps.txt" 2>&1
  cat "$E/stamps.txt"
fi
for leg in u21 u31; do python3 tools/leg.py --leg $leg --alloc pool --reps 15; done > "$E/legs.txt" 2>&1
cat "$E/legs.txt"
