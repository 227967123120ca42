#!/bin/bash
# C4 (two-word kmers + reverse complements, two arrays from the arena in two classes): threads x tile, four fresh processes per cell
O=$PWD/gpurun_out/r3c4; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2 3 4; do
  for shape in 256:512 128:512 128:384 128:640 128:768 256:768 64:256 64:512; do
    python3 tools/leg.py --leg c4 --alloc arena:0 --threads ${shape%%:*} --tile ${shape##*:} 2>> $O/err.txt | grep -v "arena map" >> $T
  done
done
cat $T
