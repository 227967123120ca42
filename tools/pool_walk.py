#!/usr/bin/env python3
"""What the striped pool finds on this box: one block of N GiB from a fresh process (KMERS_POOL_DEBUG=1 prints every unit's probes).

    KMERS_POOL_DEBUG=1 python3 tools/pool_walk.py [GiB] [search GiB]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmers_jl_amd as km

gib = int(sys.argv[1]) if len(sys.argv) > 1 else 160
ctx = km.Context(0)
if len(sys.argv) > 2:
    ctx.set_param(km._capi.PARAM_POOL_SEARCH_GIB, int(sys.argv[2]))
t0 = time.perf_counter()
p = ctx.alloc(gib << 30)
dt = time.perf_counter() - t0
info = ctx.pool_info()
chunk, classes = ctx.pool_layout(p)
runs, prev, n = [], None, 0
for c in classes + [None]:
    if c == prev:
        n += 1
    else:
        if prev is not None:
            runs.append(f"{'ABCD?'[prev]}{n}")
        prev, n = c, 1
print(f"{gib} GiB block in {dt:.2f} s; held {info['held'] / 2**30:.1f} GiB; classes {info['n_classes']} {[round(b / 2**30, 1) for b in info['class_bytes']]} GiB; "
      f"two-class {info['two_class_gbps']:.0f} one-class {info['one_class_gbps']:.0f} GB/s")
print("stripes (32 MiB each):", " ".join(runs[:40]), "..." if len(runs) > 40 else "")
t0 = time.perf_counter()
ctx.free(p)
print(f"free {time.perf_counter() - t0:.3f} s; trim released {ctx.pool_trim() / 2**30:.1f} GiB")
ctx.close()
