cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "batches" 2>&1 | tail -3
for p in 0 1 2 4 8; do python3 tools/batch_once.py --passes $p --reps 5 2>&1 | grep -v amdgpu.ids; done
python3 tools/batch_once.py --reps 5 --len 150 --reads 6700000 2>&1 | grep -v amdgpu.ids
python3 tools/batch_once.py --reps 5 --src 2 2>&1 | grep -v amdgpu.ids
python3 tools/batch_once.py --reps 5 --src 8 2>&1 | grep -v amdgpu.ids
