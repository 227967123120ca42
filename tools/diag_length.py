#!/usr/bin/env python3
"""Time per Gbase of the C2 launch (CanonicalDNAMers{31} + fx_hash, LongDNA{4}) against the length of the sequence: is there a
fixed cost per launch?  Back-to-back launches behind a warm phase, one process."""
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
LMAX, K = 8_000_000_000, 31
nw = LMAX // 16 + 2
src = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
a = torch.empty(LMAX, dtype=torch.int64, device=dev)
b = torch.empty(LMAX, dtype=torch.int64, device=dev)
flags = cap.MEM_DEVICE | cap.ASYNC
torch.cuda.synchronize()
for rnd in range(2):
    for L in (125_000_000, 250_000_000, 500_000_000, 1_000_000_000, 2_000_000_000, 4_000_000_000, 8_000_000_000):
        seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
        fn = lambda: ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, flags, C.byref(res)), "c")
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
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
        med = float(np.median(ts))
        print(f"L = {L / 1e9:5.3f} Gbase: {med:8.4f} ms per launch, {med / (L / 1e9):.4f} ms per Gbase, frac {16.5 * (L - K + 1) / med / 1e6 / 8000:.4f}", flush=True)
