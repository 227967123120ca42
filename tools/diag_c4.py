#!/usr/bin/env python3
"""C4 (FwDNAMers{63} + reverse complements, 1 Gbase LongDNA{4}, 32.5 B/kmer) reads 4.8 ms on some runs and 5.8 ms on others.
Does the relative placement of the two 16 GB output arrays decide it?  One arena, the two arrays carved at chosen offsets."""
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
L, K = 1_000_000_000, 63
n = L - K + 1
nw = L // 16 + 2
src = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
GiB = 1 << 30
arena = torch.empty(34 * GiB, dtype=torch.uint8, device=dev)
base = (arena.data_ptr() + (1 << 21) - 1) >> 21 << 21
print("arena base %#x" % base)


def timed(fn, reps=7):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    while True:  # 50 ms of the same call in front (profiles/r02_tuning.md section 1)
        fn(); t1.record(stream); t1.synchronize()
        if t0.elapsed_time(t1) > 50: break
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn(); ev[i + 1].record(stream)
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]
    return float(np.median(ts)), float(np.min(ts)), float(np.max(ts))


flags = cap.MEM_DEVICE | cap.ASYNC
OFFS = (0, 16, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, 3 << 19, 1 << 21, (1 << 21) + 2048, 5 << 20, 16 << 20, (16 << 20) + 8192, 1 << 30) if len(sys.argv) == 1 else (0, 4096)
for tile in (0,) + tuple(int(t) for t in sys.argv[1:]):
    ctx.set_param(cap.PARAM_TILE_KMERS, tile)
    for off in OFFS:
        fw = base
        rv = base + 16 * GiB + (1 << 21) + off
        med, mn, mx = timed(lambda: ctx.check(ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, fw, rv, flags, C.byref(res)), "fw"))
        print(f"tile {tile:5d} rv-fw = 16 GiB + 2 MiB + {off:>10d}: med {med:.3f} min {mn:.3f} max {mx:.3f} ms  frac {32.5 * n / med / 1e6 / 8000:.4f}", flush=True)
ctx.sync()
