#!/bin/bash
# Round-3 A/B on one box:  gpurun -- 'bash tools/r3_ab.sh'  -> gpurun_out/r3c/
O=$PWD/gpurun_out/r3c; rm -rf $O; mkdir -p $O
T=$O/times.txt
timeout 300 tools/xcd_affinity 8 > $O/xcd_affinity.txt 2>&1
OLD="env LEG_OLD_ABI=1 KMERS_HIP_LIB=$PWD/tools/libkmers_r02.so"
A="--alloc carve:150"
for rep in 1 2; do
  for leg in u31 u21; do
    $OLD python3 tools/leg.py --leg $leg $A 2>> $O/err.txt | sed 's/^/r02 /' >> $T
    python3 tools/leg.py --leg $leg $A >> $T 2>> $O/err.txt
  done
  for tile in 2048 4096; do
    $OLD python3 tools/leg.py --leg c5 $A --tile $tile 2>> $O/err.txt | sed 's/^/r02 /' >> $T
    python3 tools/leg.py --leg c5 $A --tile $tile --subtiles 1 >> $T 2>> $O/err.txt
  done
done
# C4 and C2: where the two output bases lie inside one block (same process, same physical block)
S="0:0,4096:0,8192:0,16384:0,32768:0,65536:0,0:4096,0:8192,0:16384,0:65536,4096:4096,8192:8192,65536:65536,1048576:0,2097152:0,0:2097152,1073741824:0,0:1073741824,4294967296:4294967296,8589934592:0,0:0"
python3 tools/leg.py --leg c4 --alloc carve:200 --shifts $S > $O/c4_shifts.txt 2>> $O/err.txt
python3 tools/leg.py --leg c2 --alloc carve:200 --shifts $S > $O/c2_shifts.txt 2>> $O/err.txt
for t in 256 512 768 1024; do python3 tools/leg.py --leg c4 --alloc carve:200 --tile $t >> $T 2>> $O/err.txt; done
cat $O/xcd_affinity.txt; cat $T; cat $O/c4_shifts.txt $O/c2_shifts.txt; grep -v amdgpu.ids $O/err.txt | tail -5
