cd /root/repo
R=/root/repo/gpurun_out/batch
mkdir -p $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "dense" 2>&1 | tail -3
for v in "" rgcut1 rgcut2; do
  if [ -n "$v" ]; then export KMERS_HIP_LIB=/root/repo/kmers.jl_amd/csrc/libkmers_hip_$v.so; else unset KMERS_HIP_LIB; fi
  for p in 0 2 1; do
  echo "variant '$v' passes $p"
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/trace -- python3 /root/repo/tools/batch_once.py --reps 4 --passes $p > $R/trace_run.txt 2>&1 )
  grep "ragged_kernel" $(find $R/trace -name "*kernel_stats.csv" | head -1) | cut -d, -f2-4,6-7
  rm -rf $R/trace
  done
done
unset KMERS_HIP_LIB
for p in 0 1 2 4; do python3 tools/batch_once.py --passes $p --reps 6 2>&1 | grep -v amdgpu.ids; done
