cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_pool.py -x -q -m gpu > gpurun_out/pool_tests.txt 2>&1; tail -5 gpurun_out/pool_tests.txt
{
for leg in c3 c5 f3 c4t; do
  timeout 300 python tools/leg.py --leg $leg --alloc pool 2>&1 | grep -v amdgpu.ids | tail -2
done
for shape in "256 2048" "256 3072" "128 3072" "128 4096" "256 4096"; do
  set -- $shape
  timeout 300 python tools/leg.py --leg c3 --alloc pool --threads $1 --tile $2 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python tools/leg.py --leg c3 --alloc pool --no-role 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/leg.py --leg c5 --alloc pool --no-role 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/leg.py --leg c2 --alloc pool --bases 10000000000 2>&1 | grep -v amdgpu.ids | tail -2
} > gpurun_out/pool_legs2.txt 2>&1
cat gpurun_out/pool_legs2.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; tail -15 gpurun_out/gpu_tests.txt
