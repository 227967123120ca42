#!/bin/bash
# Single-output launches through two write windows (array centred on a class boundary + split order) at the shapes that
# tools/r3_c5_shapes.sh found best, three fresh processes each, against the default launch at the bottom of the arena.
O=$PWD/gpurun_out/r3c3; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2 3; do
 for leg in c3 c5 f3 c4t; do
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 2>> $O/err.txt | grep -v "arena map" | sed "s/^/default             /" >> $T
    for shape in "128 2048" "256 3072" "128 1536" "256 4096" "64 1024" "128 1024"; do
      set -- $shape
      python3 tools/leg.py --leg $leg --alloc arenacarve:0 --straddle --split --threads $1 --tile $2 2>> $O/err.txt | grep -v "arena map\|straddle:" | sed "s/^/straddle+split      /" >> $T
    done
 done
done
cat $T; tail -3 $O/err.txt
