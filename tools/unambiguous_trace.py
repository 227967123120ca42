#!/usr/bin/env python3
"""UnambiguousDNAMers{31} over 1 Gbase (clean and p(N) = 0.04 4-bit sources) for
`rocprofv3 --kernel-trace --stats`: count / scan / emit split."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L, bits, K = 1_000_000_000, 4, 31
nw = (L * bits + 63) // 64
dev = torch.device("cuda", 0)
for amb in (0, 2621):
    buf = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, bits, amb, buf.data_ptr()), "synth")
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
    m = int(res.n_out)
    k_out = torch.empty(m, dtype=torch.int64, device=dev)
    s_out = torch.empty(m, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    for _ in range(3):
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, k_out.data_ptr(), s_out.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)), "emit")
    print(amb, m)
    del buf, k_out, s_out
