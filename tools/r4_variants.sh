#!/bin/bash
# the two UnambiguousKmers legs over variant builds of the library: bash tools/r4_variants.sh <tag> <variant> ... ("" = the product)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"; TAG="$1"; shift
E="$ROOT/gpurun_out/$TAG"; mkdir -p "$E"
cd "$ROOT"
for v in "$@"; do
  lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip_$v.so"; [ "$v" = product ] && lib="$ROOT/kmers.jl_amd/csrc/libkmers_hip.so"
  for leg in u21 u31; do
    for rep in 1 2; do
      KMERS_HIP_LIB="$lib" timeout 300 python3 tools/leg.py --leg $leg --alloc arena:0 --reps 15 2>&1 | grep "^u[23]1" | sed "s/^/$v: /"
    done
  done
done | tee "$E/variants.txt"
