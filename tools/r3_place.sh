#!/bin/bash
# The calibrated arena against plain placements:  gpurun -- 'bash tools/r3_place.sh' -> gpurun_out/r3g/
O=$PWD/gpurun_out/r3g; rm -rf $O; mkdir -p $O
T=$O/times.txt
export KMERS_ARENA_DEBUG=1
for rep in 1 2; do
  for leg in c2 c4 u31 u21; do
    python3 tools/leg.py --leg $leg --alloc arena:0 >> $T 2>> $O/err_$leg$rep.txt
    python3 tools/leg.py --leg $leg --alloc plain >> $T 2>> $O/err.txt
  done
done
cat $T; grep "arena run" $O/err_c21.txt
timeout 600 python3 -m pytest tests/test_gpu_arena.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error|Error|assert" $O/pytest.txt | tail -8
