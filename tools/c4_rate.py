#!/usr/bin/env python3
"""C4 pieces: FwDNAMers{63} over 1 Gbase LongDNA{4} with (fw, rc), fw only, and CanonicalDNAMers{63} (+ hashes): which output costs what."""
import argparse
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="fwrv,fw,canon,canonhash")
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--k", type=int, default=63)
ap.add_argument("--src", type=int, default=4)
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
L, K = 1_000_000_000, args.k
n = L - K + 1
N = (2 * K + 63) // 64
per = 64 // args.src
nw = L // per + 2
src = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, args.src, 0, src.data_ptr()), "synth")
seq = cap.Seq(src.data_ptr(), L, 0, 0, args.src, 0)
a = torch.empty(n * N, dtype=torch.int64, device=dev)
b = torch.empty(n * N, dtype=torch.int64, device=dev)
flags = cap.MEM_DEVICE | cap.ASYNC
r = args.src / 8


def timed(fn, reps):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(stream)
    while True:
        fn(); t1.record(stream); t1.synchronize()
        if t0.elapsed_time(t1) > 50: break
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record(stream)
    for i in range(reps):
        fn(); ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(reps)]))


cases = {
    "fwrv": (lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), flags, C.byref(res)), r + 16 * N),
    "fw": (lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, flags, C.byref(res)), r + 8 * N),
    "canon": (lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, flags, C.byref(res)), r + 8 * N),
    "canonhash": (lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, flags, C.byref(res)), r + 8 * N + 8),
}
for name in args.cases.split(","):
    fn, bpk = cases[name]
    ms = timed(lambda: ctx.check(fn(), name), args.reps)
    print(f"K={K} src={args.src} {name:10s} {ms:.3f} ms  {bpk:.2f} B/kmer  frac {bpk * n / ms / 1e6 / 8000:.4f}", flush=True)
ctx.sync()
