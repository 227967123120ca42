#!/bin/bash
O=$PWD/gpurun_out/r3p; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_arena.py tests/test_gpu_fuzz.py tests/test_gpu_mirror.py -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3; grep -E "^E " $O/pytest.txt | head -5
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; head -c 300 $O/bench.json; echo; grep -v amdgpu $O/bench.err | tail -3
