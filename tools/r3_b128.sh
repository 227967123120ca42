#!/bin/bash
O=$PWD/gpurun_out/r3i; rm -rf $O; mkdir -p $O
T=$O/times.txt
run() { # variant leg tile
  if [ "$1" = "b256" ]; then python3 tools/leg.py --leg $2 --alloc arena:0 --tile $3 2>> $O/err.txt | grep -v "arena map" | sed "s/^/$1 /" >> $T
  else env KMERS_HIP_LIB=$PWD/kmers.jl_amd/csrc/libkmers_hip_$1.so python3 tools/leg.py --leg $2 --alloc arena:0 --tile $3 2>> $O/err.txt | grep -v "arena map" | sed "s/^/$1 /" >> $T; fi
}
for rep in 1 2; do
  for v in b256 b128 b64; do
    for tile in 512 1024 1536 2048; do run $v c2 $tile; done
    for tile in 256 512 1024; do run $v c4 $tile; done
  done
  for tile in 512 1024 2048; do run b64 c3 $tile; run b64 c5 $tile; done
done
cat $T
timeout 600 python3 -m pytest tests/test_gpu_arena.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
