#!/bin/bash
# C5 strict (and C3): threads x tile, the one output array at the bottom of the arena in index order against the array centred on a
# class boundary with split order (two write windows).  gpurun -- 'bash tools/r3_c5_shapes.sh' -> gpurun_out/r3c5/times.txt
O=$PWD/gpurun_out/r3c5; rm -rf $O; mkdir -p $O
T=$O/times.txt
for leg in c5 c3; do
 for thr in 64 128 256; do
  for tile in 1024 1536 2048 3072 4096; do
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 --tile $tile --threads $thr 2>> $O/err.txt | grep -v "arena map" | sed "s/^/plain    thr $thr tile $tile /" >> $T
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 --straddle --split --tile $tile --threads $thr 2>> $O/err.txt | grep -v "arena map" | sed "s/^/straddle thr $thr tile $tile /" >> $T
  done
 done
done
cat $T; tail -3 $O/err.txt
