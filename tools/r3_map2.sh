#!/bin/bash
O=$PWD/gpurun_out/r3e; rm -rf $O; mkdir -p $O
timeout 300 tools/xcd_affinity 4 200 2>&1 | sed -n '/experiment 4/,/experiment 1/p' > $O/map_fill.txt
G=$((1<<30))
# both arrays moved together in 8 GiB steps
S=""; for x in $(seq 0 8 160); do S="$S,$((x*G)):$((x*G))"; done; S=${S:1}
python3 tools/leg.py --leg c4 --alloc carve:210 --shifts $S > $O/c4_map.txt 2>> $O/err.txt
# a fixed at 0, b moved away in 8 GiB steps (b's own base is 14.9 GiB behind a)
S=""; for x in $(seq 0 8 176); do S="$S,0:$((x*G))"; done; S=${S:1}
python3 tools/leg.py --leg c4 --alloc carve:210 --shifts $S > $O/c4_dist.txt 2>> $O/err.txt
python3 tools/leg.py --leg c2 --alloc carve:210 --shifts $S > $O/c2_dist.txt 2>> $O/err.txt
S=""; for x in $(seq 0 8 176); do S="$S,$((48*G)):$((x*G))"; done; S=${S:1}
python3 tools/leg.py --leg c2 --alloc carve:210 --shifts $S > $O/c2_dist48.txt 2>> $O/err.txt
cat $O/map_fill.txt; for f in c4_map c4_dist c2_dist c2_dist48; do echo "== $f"; cut -c1-80 $O/$f.txt; done
for leg in u31 u21; do python3 tools/leg.py --leg $leg --alloc carve:150; done 2>> $O/err.txt
