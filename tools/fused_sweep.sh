#!/bin/bash
# The fused consumers (XOR reducer, MinHash candidates) under the product and the KMERS_DYN_CHUNK variants, then the in-kernel
# stamps of the XOR reducer.  Run on the GPU box: bash tools/fused_sweep.sh <outdir>
OUT=${1:-gpurun_out/fused}; mkdir -p "$OUT"
C=kmers.jl_amd/csrc
for v in ""; do
  [ -f $C/libkmers_hip$v.so ] || continue
  for leg in comp8 xor; do
    echo "== lib$v $leg" >> "$OUT/legs.log"
    KMERS_HIP_LIB=$PWD/$C/libkmers_hip$v.so python3 tools/leg.py --leg $leg 2>&1 | grep " ms " >> "$OUT/legs.log"
  done
done
cat "$OUT/legs.log"
if [ -f $C/libkmers_hip_rstamps.so ]; then
  KMERS_HIP_LIB=$PWD/$C/libkmers_hip_rstamps.so RUN_STAMPS_DETAIL=1 python3 tools/run_stamps.py 2048 > "$OUT/stamps.log" 2>&1
  tail -25 "$OUT/stamps.log"
fi
