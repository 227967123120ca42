#!/usr/bin/env python3
"""Every leg of bench.py's other_configs twice, nothing else: the program rocprofv3 --pmc profiles for tools/r2_traffic.sh."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
res = cap.Result()
GOLDEN = 0x9E3779B97F4A7C15
D = cap.MEM_DEVICE


def synth(seed, n_bases, bits, amb=0):
    nw = (n_bases * bits + 63) // 64
    b = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, b.data_ptr()), "synth")
    return b


L = 1_000_000_000
a = torch.empty(2 * L, dtype=torch.int64, device=dev)
b = torch.empty(2 * L, dtype=torch.int64, device=dev)
s2 = synth(GOLDEN ^ 3, 1_250_000_000, 2)
s4 = synth(GOLDEN ^ 4, L, 4)
sa = synth(GOLDEN ^ 5, L, 4, 2621)
text = torch.tensor([65, 67, 71, 84, 97, 99, 103, 116], dtype=torch.uint8, device=dev)[torch.randint(0, 8, (L + 64,), device=dev)]
torch.cuda.synchronize()
for _ in range(2):
    seq = cap.Seq(s2.data_ptr(), 1_250_000_000, 0, 0, 2, 0)
    ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), 31, 2, a.data_ptr(), None, 0, D, C.byref(res)), "C3")
    seq = cap.Seq(s4.data_ptr(), L, 0, 0, 4, 0)
    ctx.check(ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 63, 2, a.data_ptr(), b.data_ptr(), D, C.byref(res)), "C4")
    ctx.check(ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), 21, 3, 2, a.data_ptr(), D, C.byref(res)), "C5 strict")
    seqa = cap.Seq(sa.data_ptr(), L, 0, 0, 4, 0)
    ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), 31, 1, a.data_ptr(), b.data_ptr(), L, D, C.byref(res)), "U31")
    print("U31 kept", res.n_out)
    seqt = cap.Seq(text.data_ptr(), L, 0, 0, 8, 0)
    ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seqt), 31, 2, a.data_ptr(), b.data_ptr(), 0, D, C.byref(res)), "ASCII C2")
torch.cuda.synchronize()
