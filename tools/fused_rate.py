#!/usr/bin/env python3
"""Throughput of the fused consumers (nothing materialised per kmer) on 1 Gbase LongDNA{4}:
XOR-reduce (test/benchmark.jl:9-15), MinHash sketch (docs/src/minhash.md), composition."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L = 1_000_000_000
for bits in (4, 2):
    nw = (L * bits + 63) // 64
    d = ctx.alloc(nw * 8 + 16)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 42, 0, nw, bits, 0, d), "synth")
    seq = cap.Seq(d, L, 0, 0, bits, 0)
    res = cap.Result()

    def timed(label, fn, reps=5):
        fn()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        print(f"src_bits={bits} {label:44s} {best * 1e3:8.3f} ms  {L / best / 1e9:8.1f} Gbases/s  ({L * bits / 8 / best / 1e9:6.1f} GB/s read)")

    val = C.c_uint64()
    for K in (7, 31, 63):
        timed(f"reduce_xor canonical K={K}", lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
    timed("reduce_xor forward K=31", lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 0, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
    out = np.zeros(1000, dtype=np.uint64)
    for K in (16, 31):
        timed(f"minhash sketch s=1000 K={K}", lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 1000, out.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)))
    for s_big in (100, 4000, 20000):
        big = np.zeros(s_big, dtype=np.uint64)
        timed(f"minhash sketch s={s_big} K=21", lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 21, 2, 0, s_big, big.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)))
    for K in (4, 6, 7, 8, 9, 10, 11, 12):
        counts = ctx.alloc(4 ** K * 4)
        timed(f"composition K={K}", lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, counts, cap.MEM_DEVICE, C.byref(res)))
        ctx.free(counts)
    ctx.free(d)
