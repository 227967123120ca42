#!/usr/bin/env python3
"""Does the size of the ALLOCATION an output array lives in change the rate of the C2 launch (1 Gbase LongDNA{4})?
bench.py's headline writes two fresh 8 GB allocations and reads 0.79; its 10 Gbase leg (two 80 GB allocations) 0.84."""
import ctypes as C
import os
import sys
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
flags = cap.MEM_DEVICE | cap.ASYNC


def rate(label, src, pa, pb):
    seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
    fn = lambda: ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, flags, C.byref(res)), "c")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    while True:
        fn(); t1.record(stream); t1.synchronize()
        if t0.elapsed_time(t1) > 100: break
    n = 9
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(stream)
    for i in range(n):
        fn(); ev[i + 1].record(stream)
    torch.cuda.synchronize()
    med = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))
    print(f"{label:70s} {med:.4f} ms  frac {16.5 * (L - K + 1) / med / 1e6 / 8000:.4f}", flush=True)


with torch.cuda.stream(stream):
    src = torch.zeros(nw, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
    a = torch.empty(L, dtype=torch.int64, device=dev)
    b = torch.empty(L, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
rate("two fresh 8 GB allocations (as bench.py's headline)", src, a.data_ptr(), b.data_ptr())
rate("  again", src, a.data_ptr(), b.data_ptr())
big = torch.empty(20 * L, dtype=torch.int64, device=dev)  # 160 GB
torch.cuda.synchronize()
p = big.data_ptr()
rate("both arrays inside one 160 GB allocation (offsets 0, 80 GB)", src, p, p + 80 * L)
rate("both arrays inside one 160 GB allocation (offsets 0, 8 GB)", src, p, p + 8 * L)
rate("both arrays inside it, offsets 120 GB, 150 GB", src, p + 120 * L, p + 150 * L)
rate("the two 8 GB allocations again", src, a.data_ptr(), b.data_ptr())
del big
torch.cuda.empty_cache()
c = torch.empty(L, dtype=torch.int64, device=dev)
d = torch.empty(L, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
rate("two new 8 GB allocations after freeing the 160 GB", src, c.data_ptr(), d.data_ptr())
print(torch.cuda.memory_summary(abbreviated=True)[:0])
