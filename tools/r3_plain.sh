#!/bin/bash
O=$PWD/gpurun_out/r3m; rm -rf $O; mkdir -p $O
T=$O/times.txt
timeout 600 python3 -m pytest tests/test_gpu_arena.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
gcc -std=c99 -Iinclude examples/resident_pipeline.c -Lkmers.jl_amd/csrc -lkmers_hip -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o /tmp/resident_pipeline && /tmp/resident_pipeline 1000
for rep in 1 2 3 4 5 6; do
  for tile in 1024 1536; do python3 tools/leg.py --leg c2 --alloc plain --tile $tile >> $T 2>> $O/err.txt; done
  python3 tools/leg.py --leg c4 --alloc plain >> $T 2>> $O/err.txt
done
cat $T
