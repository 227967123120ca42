#!/bin/bash
O=$PWD/gpurun_out/r3o; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2 3; do
  for leg in c5 c3; do
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 --straddle --split 2>> $O/err.txt | grep -v "arena map" >> $T
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 --straddle 2>> $O/err.txt | grep -v "arena map" >> $T
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 --split 2>> $O/err.txt | grep -v "arena map" >> $T
    python3 tools/leg.py --leg $leg --alloc arenacarve:0 2>> $O/err.txt | grep -v "arena map" >> $T
  done
done
cat $T
