#!/bin/bash
# Two-output launches with BOTH arrays in one region class (carved side by side from one block) and with plain allocations:
# index order against split order (two windows per array = four streams) over threads x tile.
O=$PWD/gpurun_out/r3sc; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2; do
 for alloc in carve:40 plain; do
  for leg in c2 c4; do
    python3 tools/leg.py --leg $leg --alloc $alloc 2>> $O/err.txt | sed "s/^/index order /" >> $T
    if [ $leg = c2 ]; then shapes="128:1536 128:2048 256:3072 256:1536 128:1024 256:1024 256:2048"; else shapes="128:512 128:1024 256:512 256:1024 256:1536 128:256"; fi
    for shape in $shapes; do
      python3 tools/leg.py --leg $leg --alloc $alloc --split --threads ${shape%%:*} --tile ${shape##*:} 2>> $O/err.txt | sed "s/^/split order /" >> $T
    done
  done
 done
done
cat $T; tail -3 $O/err.txt
