#!/bin/bash
# UnambiguousKmers after a kernel change: its GPU tests, the random geometries of tools/stress_unamb.py and the two timed legs.
#   gpurun --timeout 1500 -- 'bash tools/unamb_check.sh <tag> [stress cases]'
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
TAG="${1:-r4chk}"; CASES="${2:-80}"
E="$ROOT/gpurun_out/$TAG"; mkdir -p "$E"
cd "$ROOT"
timeout 900 python3 -m pytest tests -x -q -m gpu -k "unambiguous or skip_variant or tuple_layouts or ascii_errors or reduce_xor or spaced_skip" > "$E/tests.txt" 2>&1; echo "tests rc $?" >> "$E/tests.txt"
tail -5 "$E/tests.txt"
timeout 900 python3 tools/stress_unamb.py 1000 "$CASES" > "$E/stress.txt" 2>&1; echo "stress rc $?" >> "$E/stress.txt"
tail -3 "$E/stress.txt"
for leg in u21 u31; do timeout 300 python3 tools/leg.py --leg $leg --alloc arena:0 --reps 15; done > "$E/legs.txt" 2>&1
grep -v amdgpu.ids "$E/legs.txt"
if [ -f "$ROOT/kmers.jl_amd/csrc/libkmers_hip_stamps.so" ] && [ "${STAMPS:-1}" = 1 ]; then
  KMERS_STAMPS_LIB="$ROOT/kmers.jl_amd/csrc/libkmers_hip_stamps.so" KMERS_STAMPS_TILES=49152 KMERS_STAMPS_CASES=2 timeout 300 python3 tools/unamb_stamps.py > "$E/stamps.txt" 2>&1
  grep -v amdgpu.ids "$E/stamps.txt"
fi
