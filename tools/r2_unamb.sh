mkdir -p gpurun_out/r2e; O=$PWD/gpurun_out/r2e
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python tools/other_rates.py > $O/other_rates.txt 2>&1; grep -v amdgpu.ids $O/other_rates.txt
