#!/usr/bin/env python3
"""Does a pause in front of a timed leg change what it measures?  The C3 launch (1.25 Gbase LongDNA{2}, kmers only) right behind
a second of dense store traffic, and again after pauses of 0.2 / 1 / 3 s."""
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
L, K = 1_250_000_000, 31
nw = (L * 2 + 63) // 64
buf = torch.empty(nw + 2, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 3, 0, nw, 2, 0, buf.data_ptr()), "synth")
a = torch.empty(L, dtype=torch.int64, device=dev)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, 2, 0)
L4 = 1_000_000_000
buf4 = torch.empty(L4 // 16 + 2, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, L4 // 16, 4, 0, buf4.data_ptr()), "synth")
k4, h4 = torch.empty(L4, dtype=torch.int64, device=dev), torch.empty(L4, dtype=torch.int64, device=dev)
seq4 = cap.Seq(buf4.data_ptr(), L4, 0, 0, 4, 0)
F = cap.MEM_DEVICE | cap.ASYNC


def c3(reps=7):
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream)
        ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, F, C.byref(res))
        e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return ts


def load(seconds):
    t0 = time.time()
    while time.time() - t0 < seconds:
        for _ in range(20):
            ctx.lib.kmers_canonical(ctx.handle, C.byref(seq4), K, 2, k4.data_ptr(), h4.data_ptr(), 0, F, C.byref(res))
        torch.cuda.synchronize()


print("rested:          ", " ".join(f"{t:.3f}" for t in c3()))
for pause in (0.0, 0.2, 1.0, 3.0):
    load(1.5)
    time.sleep(pause)
    print(f"load, pause {pause:3.1f}s:", " ".join(f"{t:.3f}" for t in c3()))
