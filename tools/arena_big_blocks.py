#!/usr/bin/env python3
"""Where the arena puts the two 80 GB arrays of the 10 Gbase launch on THIS box: the measured map, the two offsets, and the share
of the arrays that lies in different classes at the same relative place (what a two-output launch writes at the same time)."""
import os
import sys

import torch  # noqa: F401  (before the library: INTEGRATION.md)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmers_jl_amd as km  # noqa: E402

ctx = km.Context(0)
ctx.arena_reserve(0)
base, gran, classes = ctx.arena_regions()
runs, start = [], 0
for i in range(1, len(classes) + 1):
    if i == len(classes) or classes[i] != classes[start]:
        runs.append(f"{chr(65 + classes[start])}{i - start}")
        start = i
print("map:", " ".join(runs))
n = 10_000_000_000 - 30
a, b = ctx.alloc(8 * n), ctx.alloc(8 * n)
src = ctx.alloc(5_000_000_016)
cls = lambda p: classes[min((p - base) // gran, len(classes) - 1)]
same = sum(cls(a + (2 * i + 1) * 8 * n // 512) == cls(b + (2 * i + 1) * 8 * n // 512) for i in range(256))
print(f"a at {(a - base) / 2**30:.1f} GiB, b at {(b - base) / 2**30:.1f} GiB, source at {(src - base) / 2**30:.1f} GiB: "
      f"{100 - same / 2.56:.0f} % of the two arrays in different classes (4 GiB map)")
print("two-stream rate of the pair, 1 GiB probes at the quarter points:",
      [round(ctx.placement_probe(a + (q * 2 * n & ~15), b + (q * 2 * n & ~15), 1 << 30)) for q in (0, 1, 2, 3)])
