cd /root/repo
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/gpu_tests.txt 2>&1; tail -40 gpurun_out/gpu_tests.txt
