#!/usr/bin/env python3
"""The single-pass UnambiguousKmers kernel in its three modes over the same 1 Gbase LongDNA{4} with p(N) = 0.04:
COUNT (stage + resolve), XOR (+ list + cut, nothing stored), EMIT (+ look-back + stores)."""
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
L = 1_000_000_000
nw = (L * 4 + 63) // 64
buf = torch.empty(nw + 2, dtype=torch.int64, device=dev); torch.cuda.synchronize()
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, 4, 2621, buf.data_ptr()), "synth")
kk = torch.empty(L // 2, dtype=torch.int64, device=dev); ss = torch.empty(L // 2, dtype=torch.int64, device=dev)
res = cap.Result(); val = C.c_uint64()


def timed(fn, reps=5):
    fn(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for K, J in ((31, 1), (21, 3)):
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
    t_count = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)))
    m = int(res.n_out)
    t_xor = timed(lambda: ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_UNAMBIGUOUS, J, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
    t_emit = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, kk.data_ptr(), ss.data_ptr(), L // 2, cap.MEM_DEVICE, C.byref(res)))
    t_k = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, kk.data_ptr(), None, L // 2, cap.MEM_DEVICE, C.byref(res)))
    OUT_TUPLES = 4
    t_t = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, kk.data_ptr(), None, L // 4, cap.MEM_DEVICE | OUT_TUPLES, C.byref(res)))
    print(f"K={K} J={J} tuples {t_t:.3f} ms", end="  ")
    print(f"K={K} J={J} kept {m}: COUNT {t_count:.3f} ms  XOR {t_xor:.3f} ms  EMIT kmers+starts {t_emit:.3f} ms  EMIT kmers only {t_k:.3f} ms")
