cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mirror.py -x -q -m gpu -k "minhash or composition or xor or fused or reducer or sketch" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "consumers" 2>&1 | tail -3
for v in "" nosel; do
  if [ -n "$v" ]; then export KMERS_HIP_LIB=/root/repo/kmers.jl_amd/csrc/libkmers_hip_$v.so; else unset KMERS_HIP_LIB; fi
  for leg in xor minhash minhash31 comp8 comp6 comp4; do
    timeout 300 python tools/leg.py --leg $leg --alloc plain 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/[$v] /"
  done
done
