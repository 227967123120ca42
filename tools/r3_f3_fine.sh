#!/bin/bash
# Two-word kmer arrays (f3) and tuple arrays (c4t) taken by role: threads x tile around the shipped shapes.
O=$PWD/gpurun_out/r3f3; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2; do
 python3 tools/leg.py --leg f3 --alloc arena:0 2>> $O/err.txt | grep -v "arena map" | sed "s/^/default /" >> $T
 for shape in 128:512 128:768 128:1024 128:1280 64:512 64:768 256:1024 256:1536 256:2048; do
   python3 tools/leg.py --leg f3 --alloc arena:0 --threads ${shape%%:*} --tile ${shape##*:} 2>> $O/err.txt | grep -v "arena map" | sed "s/^/        /" >> $T
 done
 python3 tools/leg.py --leg c4t --alloc arena:0 2>> $O/err.txt | grep -v "arena map" | sed "s/^/default /" >> $T
 for shape in 256:2048 256:2560 256:3072 256:3584 256:4096 128:2560 128:3072; do
   python3 tools/leg.py --leg c4t --alloc arena:0 --threads ${shape%%:*} --tile ${shape##*:} 2>> $O/err.txt | grep -v "arena map" | sed "s/^/        /" >> $T
 done
done
cat $T
