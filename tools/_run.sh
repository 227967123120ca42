L=kmers.jl_amd/csrc/libkmers_hip.so
timeout 300 python tools/unamb_rate.py --libs tools/libkmers_r5.so,$L --cases k31,c5,clean --reps 9 2>&1 | tail -n 6
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed" | tail -n 2
timeout 600 python tools/stress_unamb.py 2>&1 | tail -n 1
timeout 300 python tools/unamb_rate.py --libs tools/libkmers_r5.so,$L --cases k31,c5,clean --reps 9 2>&1 | tail -n 6
