cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pool.py -x -q -m gpu > gpurun_out/pool_tests.txt 2>&1; tail -5 gpurun_out/pool_tests.txt
timeout 600 python tools/n1_slices.py > gpurun_out/n1_slices.txt 2>&1; grep -v amdgpu.ids gpurun_out/n1_slices.txt
timeout 900 python bench.py > gpurun_out/bench_pool.json 2> gpurun_out/bench_pool.err; tail -5 gpurun_out/bench_pool.err; python tools/bench_summary.py gpurun_out/bench_pool.json 2>&1 | head -60
