#!/bin/bash
# Round-3 kernel sweeps on one box (one process per case, interleaved):  gpurun -- 'bash tools/r3_sweep.sh'  -> gpurun_out/r3s/
#   C5 strict: tiles x consecutive tiles per workgroup visit (the next tile's words in flight behind the stores)
#   UnambiguousKmers (framed stores): K = 31 and the C5 lattice
O=$PWD/gpurun_out/r3s; rm -rf $O; mkdir -p $O
T=$O/times.txt
timeout 900 python3 -m pytest tests/test_gpu_arena.py tests/test_gpu_comm.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt
A="--alloc arena:0"
for rep in 1 2; do
  for sub in 1 2 3 4 6; do for tile in 2048 4096; do python3 tools/leg.py --leg c5 $A --tile $tile --subtiles $sub >> $T 2>> $O/err.txt; done; done
  for leg in u31 u21 c2 c4 c4t; do python3 tools/leg.py --leg $leg $A >> $T 2>> $O/err.txt; done
done
cat $T; tail -3 $O/err.txt
