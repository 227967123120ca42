#!/bin/bash
# kernel durations of phase-cut builds of unambiguous_kernel (rocprofv3 --kernel-trace, no counters): bash tools/r4_cuts.sh <tag> <cut> <cut> ...
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
E="$ROOT/gpurun_out/$TAG"; mkdir -p "$E"
for c in "$@"; do
  lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip_ucut$c.so"; [ "$c" = 0 ] && lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip.so"
  KMERS_HIP_LIB="$lib" KERNEL=unambiguous_kernel bash "$ROOT/tools/pmc_once.sh" "$E/cut$c" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" python3 "$ROOT/tools/unamb_once.py" | sed "s/^/cut $c: /"
done | tee "$E/cuts.txt"
