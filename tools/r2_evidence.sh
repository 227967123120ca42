#!/bin/bash
# Round-2 evidence, run on the GPU box from the repo root:  gpurun -- 'bash tools/r2_evidence.sh'
#   1. the driver's command (bench.py with its own PMC child passes)          -> gpurun_out/r2ev/bench.json
#   2. the same program under rocprofv3 --kernel-trace --stats (every config)  -> gpurun_out/r2ev/trace/
# tools/r2_collect.py then writes profiles/r02_*.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
E="$ROOT/gpurun_out/r2ev"
rm -rf "$E"; mkdir -p "$E"
cd "$ROOT"
python3 bench.py > "$E/bench.json" 2> "$E/bench.err"; echo "bench rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$E/trace" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline --no-pmc > "$E/bench_under_rocprof.json" 2> "$E/trace.err"; echo "rocprof rc $?"
cd "$ROOT"
find "$E/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats.csv"
head -12 "$E/kernel_stats.csv" | cut -c1-180
cat "$E/bench.json" | head -c 6000
