#!/bin/bash
# The LDS account of composition_kernel: counter passes over kmers_composition at K = 4 and K = 8,
# 1 Gbase LongDNA{4}.   bash tools/comp_pmc.sh <outdir>
OUT=${1:-gpurun_out/comp_pmc}; mkdir -p "$OUT"
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
(cd /tmp && rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_LDS[A-Z_0-9]*" | sort -u | tr '\n' ' ') > "$ROOT/$OUT/lds_counters_available.txt"
for LEG in comp4 comp8; do
  i=0
  for SET in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES" \
             "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE SQ_LDS_MEM_VIOLATIONS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL"; do
    i=$((i+1))
    KERNEL=composition_kernel bash "$ROOT/tools/pmc_once.sh" "$ROOT/$OUT/${LEG}_p$i" "$SET" python3 "$ROOT/tools/leg.py" --leg $LEG --alloc plain --reps 2 2>&1 | tail -1 | sed "s/^/$LEG: /" | tee -a "$ROOT/$OUT/pmc.txt"
    tail -2 "$ROOT/$OUT/${LEG}_p$i/run.txt" | grep -i "error\|invalid\|not found" | head -2 | sed "s/^/$LEG p$i: /" >> "$ROOT/$OUT/pmc.txt"
  done
done
cat "$ROOT/$OUT/lds_counters_available.txt"
