#!/usr/bin/env python3
"""One MinHash sketch (s = 1000, K = 16, 1 Gbase 4-bit source) for `rocprofv3 --kernel-trace --stats`:
shows how the time splits between candidate kernels and prune kernels."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L, bits, K = 1_000_000_000, 4, 16
nw = (L * bits + 63) // 64
d = ctx.alloc((nw + 2) * 8)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, bits, 0, d), "synth")
seq = cap.Seq(d, L, 0, 0, bits, 0)
res = cap.Result()
out = np.zeros(1000, dtype=np.uint64)
for _ in range(3):
    ctx.check(ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 1000, out.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)), "minhash")
print(res.n_out, hex(int(out[0])), hex(int(out[-1])))
