timeout 300 python tools/ascii_rate.py '' 1024,1536,2048 2>&1 | grep 4-bit
timeout 300 python tools/sweep.py --tiles 1024,1536,2048 --rounds 7 2>&1 | tail -n 3
timeout 300 python tools/ascii_rate.py '' 1024,1536,2048 2>&1 | grep 4-bit
