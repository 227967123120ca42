cd /root/repo
mkdir -p gpurun_out
KMERS_POOL_DEBUG=1 timeout 300 python tools/pool_walk.py 200 0 > gpurun_out/pool_walk2.txt 2>&1
tail -12 gpurun_out/pool_walk2.txt
timeout 900 python -m pytest tests/test_gpu_pool.py -x -q -m gpu > gpurun_out/pool_tests.txt 2>&1; tail -30 gpurun_out/pool_tests.txt
{
for leg in c2 c3 c4 c5 f1 u31 u21 f3 c63h; do
  timeout 300 python tools/leg.py --leg $leg --alloc pool 2>&1 | grep -v amdgpu.ids | tail -2
done
timeout 300 python tools/leg.py --leg c2 --alloc plain 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/leg.py --leg c3 --alloc plain 2>&1 | grep -v amdgpu.ids | tail -1
} > gpurun_out/pool_legs.txt 2>&1
cat gpurun_out/pool_legs.txt
