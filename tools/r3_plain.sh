#!/bin/bash
# The arena against plain allocations, fresh processes (profiles/r03_tuning.md section 2):   gpurun -- 'bash tools/r3_plain.sh'  -> gpurun_out/r3m/
O=$PWD/gpurun_out/r3m; rm -rf $O; mkdir -p $O
T=$O/times.txt
gcc -std=c99 -Iinclude examples/resident_pipeline.c -Lkmers.jl_amd/csrc -lkmers_hip -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o /tmp/resident_pipeline && /tmp/resident_pipeline 1000
export KMERS_ARENA_DEBUG=1
for rep in 1 2 3 4; do
  for leg in c2 c4 u31 u21; do python3 tools/leg.py --leg $leg --alloc arena:0 >> $T 2>> $O/map_$leg$rep.txt; done
  for tile in 1024 1536; do python3 tools/leg.py --leg c2 --alloc plain --tile $tile >> $T 2>> $O/err.txt; done
  python3 tools/leg.py --leg c4 --alloc plain >> $T 2>> $O/err.txt
done
cat $T; grep "arena run" $O/map_c21.txt | cut -c1-110
