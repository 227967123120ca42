#!/usr/bin/env python3
"""One fresh process per case: the C2 launch with its two 8 GB outputs carved from ONE allocation of the given size (GB), or
(size 0) two separate 8 GB allocations; 'prefree N': N GB allocated and freed first, then two separate 8 GB allocations."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
mode, size = sys.argv[1], int(sys.argv[2])
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
L, K = 1_000_000_000, 31
nw = L // 16 + 2
flags = cap.MEM_DEVICE | cap.ASYNC
src = torch.zeros(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
if mode == "arena" and size:
    arena = torch.empty(size * (1 << 30) // 8, dtype=torch.int64, device=dev)
    pa, pb = arena.data_ptr(), arena.data_ptr() + 8 * L + (1 << 21)
else:
    if mode == "prefree":
        tmp = torch.empty(size * (1 << 30) // 8, dtype=torch.int64, device=dev)
        del tmp
        torch.cuda.empty_cache()
    a = torch.empty(L, dtype=torch.int64, device=dev)
    b = torch.empty(L, dtype=torch.int64, device=dev)
    pa, pb = a.data_ptr(), b.data_ptr()
torch.cuda.synchronize()
seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
fn = lambda: ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, flags, C.byref(res)), "c")
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(stream)
while True:
    fn(); t1.record(stream); t1.synchronize()
    if t0.elapsed_time(t1) > 300: break
n = 9
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record(stream)
for i in range(n):
    fn(); ev[i + 1].record(stream)
torch.cuda.synchronize()
med = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))
print(f"{mode:8s} {size:4d} GB: {med:.4f} ms  frac {16.5 * (L - K + 1) / med / 1e6 / 8000:.4f}   a at {pa:#x}", flush=True)
