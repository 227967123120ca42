#!/usr/bin/env python3
"""How the store rate of two streams depends on WHICH two region classes they lie in (round 5): the class pool's own probes
(KMERS_POOL_DEBUG=1: every new 1 GiB handle is timed beside the representative of every class found so far, two 1 GiB store
streams) collected over a walk of 200 GiB and tabulated by (class of the handle, class of the representative).
    KMERS_POOL_DEBUG=1 python3 tools/pool_pairs.py 2> walk.txt; python3 tools/pool_pairs.py --table walk.txt"""
import collections
import os
import re
import sys
if len(sys.argv) > 2 and sys.argv[1] == "--table":
    rows = collections.defaultdict(list)
    for line in open(sys.argv[2]):
        m = re.match(r"pool chunk\s+(\d+): ([ABC?])(.*?)\((?:one class [\d.]+ us, two [\d.]+ us;)(.*)\)", line)
        if not m or "representative" in m.group(3):
            continue
        for rep, us in re.findall(r"([ABC]):?\s*([\d.]+)", m.group(4)):
            rows[m.group(2), rep].append(float(us))
    print("class of the handle, class of the representative beside it: probes, median us for 2 x 1 GiB, TB/s")
    for (c, r), v in sorted(rows.items()):
        v.sort()
        med = v[len(v) // 2]
        print(f"  {c} beside {r}: {len(v):4d} probes, median {med:7.1f} us = {2 * 2**30 / med / 1e6:5.2f} TB/s   (min {v[0]:.1f}, max {v[-1]:.1f})")
    sys.exit(0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import kmers_jl_amd as km
ctx = km.Context(0)
p = ctx.alloc(int(os.environ.get("WALK_GIB", "200")) << 30)
info = ctx.pool_info()
print(f"held {info['held'] / 2**30:.0f} GiB, classes {info['n_classes']} {[round(b / 2**30) for b in info['class_bytes']]} GiB")
ctx.free(p)
