cd /root/repo
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; tail -5 gpurun_out/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
