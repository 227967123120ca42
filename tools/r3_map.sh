#!/bin/bash
# Where in a large block do the output arrays lie, and does it matter?   gpurun -- 'bash tools/r3_map.sh'  -> gpurun_out/r3d/
O=$PWD/gpurun_out/r3d; rm -rf $O; mkdir -p $O
timeout 300 tools/xcd_affinity 4 200 2>&1 | grep -A3 "experiment 4" > $O/map_fill.txt
G=$((1<<30)); S=""; for x in 0 16 32 48 64 80 96 112 128 144 160; do S="$S,$((x*G)):$((x*G))"; done; S=${S:1}
python3 tools/leg.py --leg c4 --alloc carve:200 --shifts $S > $O/c4_map.txt 2>> $O/err.txt
python3 tools/leg.py --leg c2 --alloc carve:200 --shifts $S > $O/c2_map.txt 2>> $O/err.txt
python3 tools/leg.py --leg c4 --alloc plain >> $O/c4_map.txt 2>> $O/err.txt
python3 tools/leg.py --leg c4 --alloc arena:0 >> $O/c4_map.txt 2>> $O/err.txt
python3 tools/leg.py --leg c4 --alloc arena:100 >> $O/c4_map.txt 2>> $O/err.txt
cat $O/map_fill.txt $O/c4_map.txt $O/c2_map.txt
timeout 1200 python3 -m pytest tests/test_gpu_arena.py tests/test_gpu_comm.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 6000 $O/bench.json; tail -5 $O/bench.err
