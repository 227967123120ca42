#!/bin/bash
# legs of tools/leg.py over builds of the library, same box, two fresh processes per cell: bash tools/variants.sh <tag> "<legs>" <variant> ...
#   variant = product | <name> of a kmers.jl_amd/csrc/libkmers_hip_<name>.so (python -m kmers_jl_amd.build variant <name> -DFLAG ... [unit.hip ...])
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; LEGS="$2"; shift 2
E="$ROOT/gpurun_out/$TAG"; mkdir -p "$E"
cd "$ROOT"
for v in "$@"; do
  lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip_$v.so"; [ "$v" = product ] && lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip.so"
  for leg in $LEGS; do
    for rep in 1 2; do
      KMERS_HIP_LIB="$lib" timeout 300 python3 tools/leg.py --leg $leg --alloc pool --reps 15 2>&1 | grep -v "amdgpu.ids" | tail -2 | tr '\n' ' ' | sed "s/^/$v: /"; echo
    done
  done
done | tee "$E/legs_ab.txt"
