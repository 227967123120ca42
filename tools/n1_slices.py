#!/usr/bin/env python3
"""Why does the 10 Gbase launch (north star, one GPU) run 2.5-4 % below the 1 Gbase launch per base?  

One process, outputs from the class pool (every relative position of the two 80 GB arrays in different region classes):
  a. the whole 10 Gbase in ONE launch
  b. the same work as TEN launches of 1 Gbase into consecutive slices of the same arrays, back to back: each slice timed
  c. the ten slices again with 100 ms of idle device between them
  d. ONE slice launched ten times in a row (1 Gbase working set, sustained for as long as the big launch)
If (b) sums to (a): the size of the grid is not it.  If the slices of (b) slow down in the course of the 24 ms while (c) does not:
the device slows under sustained load (clocks / power), not the kernel.  If particular slices are slow in (b) AND (c): their memory.

    python3 tools/n1_slices.py [--gbases 10]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import kmers_jl_amd as km

ap = argparse.ArgumentParser()
ap.add_argument("--gbases", type=int, default=10)
ap.add_argument("--alloc", default="pool")
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
K, S = 31, args.gbases
L1 = 1_000_000_000
L = S * L1
n = L - K + 1
nw = (L * 4 + 63) // 64
pa, pb, psrc = ctx.alloc(8 * n), ctx.alloc(8 * n), ctx.alloc(8 * (nw + 2))
info = ctx.pool_info()
runs = lambda p: "".join("ABC?"[c] for c in ctx.pool_layout(p)[1])
print(f"pool held {info['held'] / 2**30:.0f} GiB, in use {info['in_use'] / 2**30:.0f}; a {runs(pa)}\n{' ' * 38}b {runs(pb)}", flush=True)
# (round 6) the handles under every 1 Gbase slice of the two arrays: what a slow slice lies in
la, lb = runs(pa), runs(pb)
GiB = 1 << 30
for i in range(S):
    lo, hi = 8 * i * L1 // GiB, min(len(la) - 1, (8 * (i + 1) * L1 - 1) // GiB)
    print(f"slice {i}: a {la[lo:hi + 1]}  b {lb[lo:hi + 1]}  differ at {sum(x != y for x, y in zip(la[lo:hi + 1], lb[lo:hi + 1]))}/{hi - lo + 1} handles", flush=True)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 0x9E3779B97F4A7C15 ^ 10, 0, nw, 4, 0, psrc), "synth")
res = cap.Result()
ASYNC = cap.MEM_DEVICE | cap.ASYNC
whole = cap.Seq(psrc, L, 0, 0, 4, 0)


def launch_whole():
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(whole), K, 2, pa, pb, 0, ASYNC, C.byref(res)) == 0, ctx.last_error()


def launch_slice(i):
    first = i * L1  # a view of 1 Gbase (+ K - 1 symbols of the next slice): elements [first, first + L1)
    nb = min(L1 + K - 1, L - first)
    seq = cap.Seq(psrc, nb, first, 0, 4, 0)
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa + 8 * first, pb + 8 * first, 0, ASYNC, C.byref(res)) == 0, ctx.last_error()


def timed(fns, gap_s=0.0):
    out = []
    with torch.cuda.stream(stream):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in fns]
        for f, (e0, e1) in zip(fns, ev):
            e0.record(stream)
            f()
            e1.record(stream)
            if gap_s:
                torch.cuda.synchronize()
                time.sleep(gap_s)
        torch.cuda.synchronize()
        out = [e0.elapsed_time(e1) for e0, e1 in ev]
    return out


with torch.cuda.stream(stream):  # wake the device
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        launch_slice(0)
        torch.cuda.synchronize()
frac = lambda ms, bases: 16.5 * bases / ms / 1e6 / 8000
a = timed([launch_whole] * 5)
print("a. one launch of %d Gbase: %s ms -> frac %.4f" % (S, " ".join(f"{t:.2f}" for t in a), frac(float(np.median(a)), L)), flush=True)
for rep in range(2):
    b = timed([(lambda i=i: launch_slice(i)) for i in range(S)])
    print("b. slices back to back   : %s | sum %.2f ms -> frac %.4f" % (" ".join(f"{t:.3f}" for t in b), sum(b), frac(sum(b), L)), flush=True)
c = timed([(lambda i=i: launch_slice(i)) for i in range(S)], gap_s=0.1)
print("c. slices, 100 ms apart  : %s | sum %.2f ms -> frac %.4f" % (" ".join(f"{t:.3f}" for t in c), sum(c), frac(sum(c), L)), flush=True)
for s_i in (0, S // 2):
    d = timed([(lambda: launch_slice(s_i))] * S)
    print("d. slice %d ten times     : %s | sum %.2f ms -> frac %.4f" % (s_i, " ".join(f"{t:.3f}" for t in d), sum(d), frac(sum(d), L)), flush=True)
e = timed([launch_whole] * 3)
print("a'. one launch again     : %s ms" % " ".join(f"{t:.2f}" for t in e), flush=True)
for p in (pa, pb, psrc):
    ctx.free(p)
