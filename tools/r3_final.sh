#!/bin/bash
O=$PWD/gpurun_out/r3p; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_arena.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -2
for rep in 1 2; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench$rep.json 2> $O/bench$rep.err; echo "bench rc $?"
done
python3 - <<'PY'
import json
for r in (1, 2):
    d = json.loads(open(f"gpurun_out/r3p/bench{r}.json").read().strip().splitlines()[0])
    print(r, d["value"], d["roofline"]["frac"], d["roofline"].get("plain_alloc", {}).get("frac"), d["config"]["arena_region_map"])
    for k, v in d.get("other_configs", {}).items():
        print("    ", k[:50], v.get("frac_of_8TBps"), v.get("kernel_ms", v.get("ms")), v.get("verified"))
PY
