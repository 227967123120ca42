#!/usr/bin/env python3
"""How long does a fresh box take to reach its steady rate?  The C2 launch (CanonicalDNAMers{31} + fx_hash, 1 Gbase LongDNA{4})
back to back from process start, kernel time per launch against the time since the first launch; then the same after
idle gaps of 1, 5 and 20 s."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
L, K = 1_000_000_000, 31
nw = L // 16 + 2
src = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
a = torch.empty(L, dtype=torch.int64, device=dev)
b = torch.empty(L, dtype=torch.int64, device=dev)
flags = cap.MEM_DEVICE | cap.ASYNC
torch.cuda.synchronize()


def burst(seconds, label):
    t_start = time.perf_counter()
    rows = []
    while time.perf_counter() - t_start < seconds:
        n = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        t_rel = time.perf_counter() - t_start
        ev[0].record(stream)
        for i in range(n):
            ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, flags, C.byref(res)), "c")
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        rows.append((t_rel, float(np.median(ts)), min(ts), max(ts)))
    marks = [0, 1, 2, 3, 5, 8, 12, 16, 20, 30, 40, 50, 60, 80, 100, 120]
    out = []
    for m in marks:
        near = [r for r in rows if r[0] >= m]
        if near:
            r = near[0]
            out.append(f"t={r[0]:6.2f}s med {r[1]:.3f} (min {r[2]:.3f} max {r[3]:.3f}) frac {16.5 * (L - K + 1) / r[1] / 1e6 / 8000:.3f}")
    print(label)
    for o in out:
        print("   ", o, flush=True)


burst(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, "from process start (fresh box):")
for gap in (1.0, 5.0, 20.0):
    time.sleep(gap)
    burst(6.0, f"after {gap:.0f} s idle:")
