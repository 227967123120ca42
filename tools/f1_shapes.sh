#!/bin/bash
# f1 (the headline launch from ASCII text) over launch shapes, outputs from the class pool: bash tools/f1_shapes.sh <outdir>
OUT=${1:-gpurun_out/f1shapes}; mkdir -p "$OUT"
for leg in f1 c2; do
for shape in "0 0" "128 1536" "128 1024" "128 2048" "256 1024" "256 1536" "256 2048" "64 1024" "128 3072"; do
  set -- $shape
  echo "== $leg threads $1 tile $2" >> "$OUT/shapes.log"
  python3 tools/leg.py --leg $leg --alloc pool --reps 9 --threads $1 --tile $2 2>&1 | grep " ms " >> "$OUT/shapes.log"
done
done
cat "$OUT/shapes.log"
