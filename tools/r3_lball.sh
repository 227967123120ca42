#!/bin/bash
O=$PWD/gpurun_out/r3n; rm -rf $O; mkdir -p $O
T=$O/times.txt
V="env KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_lball.so"
for rep in 1 2 3; do
  for leg in u31 u21; do
    python3 tools/leg.py --leg $leg --alloc arena:0 2>> $O/err.txt | grep -v "arena map" >> $T
    $V python3 tools/leg.py --leg $leg --alloc arena:0 2>> $O/err.txt | grep -v "arena map" | sed 's/^/lball /' >> $T
  done
done
cat $T
$V timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k unambiguous 2>&1 | grep -E "passed|failed|error" | tail -2
