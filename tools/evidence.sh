#!/bin/bash
# Everything committed under profiles/<round>_* (KMERS_ROUND, default r04) comes from this script, run on the GPU box from the repo root:
#   gpurun --timeout 1700 -- 'bash tools/evidence.sh'      then, here:   python tools/evidence.py
#   1. the driver's command (bench.py with its own PMC child passes)                          -> bench.json
#   2. the same program under rocprofv3 --kernel-trace --stats, headline leg only (--no-other-configs: the other legs launch the
#      same kernel at other sizes and placements and would share its row)                      -> kernel_stats.csv
#   3. one --kernel-trace --stats pass PER LEG (tools/leg.py --no-calibrate: the launcher's table shape only, so that a leg is
#      one row of its own file and not two)                                                      -> kernel_stats_<leg>.csv
#   4. one FETCH_SIZE and one WRITE_SIZE pass per leg (separate passes: TCC slots)             -> pmc_<counter>_<leg>/
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
RND="${KMERS_ROUND:-r04}"
E="$ROOT/gpurun_out/${RND}ev"
rm -rf "$E"; mkdir -p "$E"
cd "$ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$E/bench.json" 2> "$E/bench.err"; echo "bench rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$E/trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc --no-shape-calibration > "$E/bench_under_rocprof.json" 2> "$E/trace.err"; echo "rocprof rc $?"
find "$E/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats.csv"
for leg in c2 c3 c4 c5 f1 f3 u31 u21 xor minhash; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$E/stats_$leg" -- python3 "$ROOT/tools/leg.py" --leg $leg --alloc arena:0 --reps 20 --no-calibrate > "$E/stats_$leg.txt" 2>&1
  find "$E/stats_$leg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats_$leg.csv"
  rm -rf "$E/stats_$leg"
done
for leg in c3 c4 c5 f3 u31 u21; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$E/pmc_${c}_$leg" -- python3 "$ROOT/tools/leg.py" --leg $leg --alloc arena:0 --once --no-calibrate > "$E/pmc_${c}_$leg.txt" 2>&1
  done
done
cd "$ROOT"
rm -rf "$E/trace"
head -c 3000 "$E/bench.json"; echo; tail -3 "$E/bench.err"
for f in "$E"/stats_*.txt; do tail -1 "$f"; done
