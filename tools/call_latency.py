#!/usr/bin/env python3
"""Fixed cost of one call through the C ABI on short sequences (FASTA-record sized inputs):
microseconds per synchronous call for a few entry points, host and device pointers."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
from oracle import pyoracle
cap = km._capi
ctx = km.Context(0)
orc = pyoracle.get()
res = cap.Result()
for L in (1_000, 100_000):
    K = 31
    words = orc.synth_words(1, 0, L // 16 + 2, 4)
    n = L - K + 1
    out_k, out_h = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    d_w = ctx.alloc(words.nbytes)
    ctx.h2d(d_w, words)
    d_k, d_h = ctx.alloc(n * 8), ctx.alloc(n * 8)
    val = C.c_uint64()
    sk = np.zeros(100, np.uint64)
    cases = {
        "canonical+hash, host pointers": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)), K, 2, out_k.ctypes.data, out_h.ctypes.data, 0, cap.MEM_HOST, C.byref(res)),
        "canonical+hash, device pointers": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(cap.Seq(d_w, L, 0, 0, 4, 0)), K, 2, d_k, d_h, 0, cap.MEM_DEVICE, C.byref(res)),
        "reduce_xor, device source": lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(cap.Seq(d_w, L, 0, 0, 4, 0)), K, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)),
        "minhash s=100, device source": lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(cap.Seq(d_w, L, 0, 0, 4, 0)), K, 2, 0, 100, sk.ctypes.data, cap.MEM_DEVICE, C.byref(res)),
        "unambiguous count, device source": lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(cap.Seq(d_w, L, 0, 0, 4, 0)), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)),
    }
    for name, fn in cases.items():
        for _ in range(20):
            assert fn() == 0
        t0 = time.perf_counter()
        reps = 300
        for _ in range(reps):
            fn()
        dt = (time.perf_counter() - t0) / reps
        print(f"L={L:7d}  {name:36s} {dt * 1e6:8.1f} us per call")
    for d in (d_w, d_k, d_h):
        ctx.free(d)
