#!/bin/bash
# Collects the rocprofv3 evidence for the headline kernel on the GPU box:
#   1. kernel trace + stats of `bench.py` (per-kernel average duration)
#   2. PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs: TCC slots) + SQ/GRBM counters
# Usage (from the repo root on the GPU box):  bash tools/profile.sh [extra bench args]
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="$ROOT/bench.py"
ARGS="--no-cpu-baseline --no-other-configs --no-pmc $*"  # --no-pmc: never a nested rocprofv3 under this one
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$BENCH" --steps 10 --warmup 2 $ARGS > "$OUT/trace_bench.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 "$BENCH" --steps 3 --warmup 1 $ARGS > "$OUT/pmc_fetch_bench.json" 2> "$OUT/pmc_fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 "$BENCH" --steps 3 --warmup 1 $ARGS > "$OUT/pmc_write_bench.json" 2> "$OUT/pmc_write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_sq" -- python3 "$BENCH" --steps 3 --warmup 1 $ARGS > "$OUT/pmc_sq_bench.json" 2> "$OUT/pmc_sq.err"
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
ls -R "$OUT" | head -50
