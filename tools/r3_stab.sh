#!/bin/bash
O=$PWD/gpurun_out/r3k; rm -rf $O; mkdir -p $O
T=$O/times.txt
timeout 900 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_arena.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
export KMERS_ARENA_DEBUG=1
for rep in 1 2 3 4; do
  for leg in c2 c4 u31; do
    python3 tools/leg.py --leg $leg --alloc arena:0 >> $T 2>> $O/map_$leg$rep.txt
  done
  python3 tools/leg.py --leg c2 --alloc plain >> $T 2>> $O/err.txt
done
cat $T; for f in $O/map_c2*.txt; do grep "arena run" $f | cut -c1-60; echo; done
