#!/bin/bash
O=$PWD/gpurun_out/r3j; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2; do
 for thr in 128 256; do
  for tile in 1024 1536 2048 3072; do
    python3 tools/leg.py --leg c2 --alloc arena:0 --tile $tile --threads $thr 2>> $O/err.txt | grep -v "arena map" >> $T
    python3 tools/leg.py --leg c2 --alloc carve:40 --tile $tile --threads $thr --shifts 0:0 2>> $O/err.txt | sed "s/^/same-class tile $tile thr $thr /" >> $T
  done
  for tile in 512 1024 1536; do
    python3 tools/leg.py --leg c4 --alloc arena:0 --tile $tile --threads $thr 2>> $O/err.txt | grep -v "arena map" >> $T
    python3 tools/leg.py --leg c4 --alloc carve:40 --tile $tile --threads $thr --shifts 0:0 2>> $O/err.txt | sed "s/^/same-class tile $tile thr $thr /" >> $T
  done
 done
 for thr in 128 256; do for tile in 2048 4096; do
    python3 tools/leg.py --leg c3 --alloc arena:0 --tile $tile --threads $thr 2>> $O/err.txt | grep -v "arena map" >> $T
 done; done
 for leg in u31 u21; do python3 tools/leg.py --leg $leg --alloc arena:0 2>> $O/err.txt | grep -v "arena map" >> $T; done
done
cat $T
