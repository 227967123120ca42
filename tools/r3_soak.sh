#!/bin/bash
O=$PWD/gpurun_out/r3l; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|error" $O/pytest.txt | tail -3
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python3 tools/stress_unamb.py 5000 400 > $O/stress_unamb.txt 2>&1; tail -2 $O/stress_unamb.txt
timeout 900 python3 tools/stress_fuzz.py 300 340 > $O/stress_fuzz.txt 2>&1; tail -2 $O/stress_fuzz.txt
