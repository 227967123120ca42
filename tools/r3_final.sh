#!/bin/bash
O=$PWD/gpurun_out/r3p; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_arena.py tests/test_bench_contract.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
KMERS_ARENA_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 600 $O/bench.json; echo; grep "arena run" $O/bench.err | cut -c1-120
