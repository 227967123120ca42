#!/bin/bash
# C2 (the headline launch, two arrays from the arena in two classes): threads x tile around the shipped 128 x 1536, fresh processes
O=$PWD/gpurun_out/r3c2s; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2 3 4; do
  for shape in ${SHAPES:-128:1536 128:1280 128:1792 64:512 64:768 64:1024 256:1536 256:2560 128:1024 64:1280}; do
    python3 tools/leg.py --leg ${LEG:-c2} --alloc arena:0 --threads ${shape%%:*} --tile ${shape##*:} 2>> $O/err.txt | grep -v "arena map" >> $T
  done
done
cat $T
