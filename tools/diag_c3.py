#!/usr/bin/env python3
"""Why does bench.py's other_configs leg time the C3 shard slower than tools/sweep.py?  Same launch, timed the two ways."""
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


def timed(fn, reps=7, presync=True):
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if presync:
            torch.cuda.synchronize()
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


def run(label, L, bits, K, in_stream_ctx):
    nw = (L * bits + 63) // 64
    def body():
        buf = torch.empty(nw + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 3, 0, nw, bits, 0, buf.data_ptr()), "synth")
        a = torch.empty(L, dtype=torch.int64, device=dev)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
        torch.cuda.synchronize()
        for name, flags in (("sync", cap.MEM_DEVICE), ("async", cap.MEM_DEVICE | cap.ASYNC)):
            for presync in (False, True):
                med, mn = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, flags, C.byref(res)), presync=presync)
                print(f"{label:30s} {name:5s} presync={presync!s:5s} med {med:.4f} min {mn:.4f} ms  frac {8.25 * (L - K + 1) / med / 1e6 / 8000:.4f}", flush=True)
        ctx.sync()
    if in_stream_ctx:
        with torch.cuda.stream(stream):
            body()
    else:
        body()


run("fresh, default torch stream", 1_250_000_000, 2, 31, False)
run("fresh, lib stream ctx", 1_250_000_000, 2, 31, True)
big = torch.empty(20_000_000_000, dtype=torch.int64, device=dev)  # 160 GB
big[:1 << 20].zero_()
del big
torch.cuda.empty_cache()
run("after 160 GB alloc/free", 1_250_000_000, 2, 31, True)
run("L = 1.0e9", 1_000_000_000, 2, 31, True)
