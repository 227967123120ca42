cd /root/repo
R=/root/repo/gpurun_out/batch
mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "batches" 2>&1 | tail -3
for p in 0 4; do python3 tools/batch_once.py --passes $p --reps 8 2>&1 | grep -v amdgpu.ids; done
python3 tools/batch_once.py --reps 6 --src 2 2>&1 | grep -v amdgpu.ids
python3 tools/batch_once.py --reps 6 --src 8 2>&1 | grep -v amdgpu.ids
python3 tools/batch_once.py --reps 6 --len 150 --reads 6700000 2>&1 | grep -v amdgpu.ids
python3 tools/batch_once.py --reps 6 --len 36 --reads 30000000 --src 2 2>&1 | grep -v amdgpu.ids
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- python3 /root/repo/tools/batch_once.py --reps 4 > $R/trace_run.txt 2>&1 )
cut -d, -f1-4 $(find $R/trace -name "*kernel_stats.csv" | head -1) | grep -v "at::\|probe\|rocclr" | head -12
rm -rf $R/trace
