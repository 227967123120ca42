#!/bin/bash
# C5 strict with its array taken by role (across a class boundary, split order): tiles beyond what MAX_TILE_BITS = 32768 allows
# (variant builds: python -m kmers_jl_amd.build variant tb64 -DKMERS_MAX_TILE_BITS=65536, tb128 ...=131072)
O=$PWD/gpurun_out/r3c5f; rm -rf $O; mkdir -p $O
T=$O/times.txt
for rep in 1 2 3; do
 python3 tools/leg.py --leg c5 --alloc arena:0 --threads 256 --tile 5120 2>> $O/err.txt | grep -v "arena map" | sed "s/^/product 32768 /" >> $T
 for tile in 5120 6144 7680 10240; do
   KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_tb64.so python3 tools/leg.py --leg c5 --alloc arena:0 --threads 256 --tile $tile 2>> $O/err.txt | grep -v "arena map" | sed "s/^/tb64          /" >> $T
 done
 for tile in 10240 15360 20480; do
   KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_tb128.so python3 tools/leg.py --leg c5 --alloc arena:0 --threads 256 --tile $tile 2>> $O/err.txt | grep -v "arena map" | sed "s/^/tb128         /" >> $T
 done
 for tile in 2048 3072 4096 5120; do
   KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_tb128.so python3 tools/leg.py --leg c3 --alloc arena:0 --threads 256 --tile $tile 2>> $O/err.txt | grep -v "arena map" | sed "s/^/tb128         /" >> $T
   KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_tb128.so python3 tools/leg.py --leg c3 --alloc arena:0 --threads 128 --tile $tile 2>> $O/err.txt | grep -v "arena map" | sed "s/^/tb128         /" >> $T
 done
done
cat $T
