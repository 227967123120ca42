#!/bin/bash
# The library must not contain FLAT memory instructions: on gfx950 a FLAT load whose address is
# selected between the LDS aperture and HBM returned wrong data for whole wavefronts
# ("Batches of records").  Compiles the device code to assembly and counts them.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${TMPDIR:-/tmp}/kmers_isa_$$"
mkdir -p "$OUT"
n=0
for f in "$ROOT"/kmers.jl_amd/csrc/*_api.hip; do   # every translation unit of the library
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Wno-array-bounds -o "$OUT/$(basename "$f").s" "$f" 2>/dev/null
  c=$(grep -c "flat_load\|flat_store\|flat_atomic" "$OUT/$(basename "$f").s" || true)
  n=$((n + c))
done
rm -rf "$OUT"
echo "FLAT memory instructions: $n"
test "$n" = "0"
