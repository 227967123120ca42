#!/bin/bash
# Everything committed under profiles/ comes from this script, run on the GPU box from the repo root:
#   rm -rf gpurun_out/prof gpurun_out/evidence   (on the build host: gpurun merges files back, it never deletes stale ones)
#   gpurun -- 'bash tools/refresh_evidence.sh'
#   python tools/collect_evidence.py
# It writes under gpurun_out/evidence/ (merged back by gpurun); collect_evidence.py turns that into profiles/r01_*.
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
E="$ROOT/gpurun_out/evidence"
rm -rf "$ROOT/gpurun_out/prof" "$E"
mkdir -p "$E"
cd "$ROOT"
python3 bench.py > "$E/bench.json" 2> "$E/bench.err"
bash tools/profile.sh > "$E/profile.log" 2>&1          # rocprofv3 trace + PMC passes -> gpurun_out/prof
python3 tools/other_rates.py > "$E/other_rates.txt" 2>&1
python3 tools/fused_rate.py > "$E/fused_rates.txt" 2>&1
python3 tools/batch_rate.py > "$E/batch_rates.txt" 2>&1
python3 tools/sketch_batch_rate.py > "$E/sketch_batch_rates.txt" 2>&1
python3 tools/sketch_lowcomplexity.py > "$E/sketch_lowcomplexity.txt" 2>&1
python3 tools/call_latency.py > "$E/call_latency.txt" 2>&1
python3 tools/reference_benchmark.py > "$E/reference_benchmark_10m.txt" 2>&1
python3 tools/reference_benchmark.py --n 1000000000 > "$E/reference_benchmark_1g.txt" 2>&1
ls -la "$E"
