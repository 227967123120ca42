#!/usr/bin/env python3
"""UnambiguousKmers on the C5 lattice (K = 21, stride 3) and at K = 31, two launches each and nothing else: the program the
PMC passes of tools/unamb_account.sh profile (KMERS_HIP_LIB selects the build: the product or a -DKMERS_UCUT=n phase cut)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmers_jl_amd as km

cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
L = int(os.environ.get("UNAMB_BASES", "1000000000"))
nw = (L * 4 + 63) // 64
src = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 0x9E3779B97F4A7C15 ^ 5, 0, nw, 4, 2621, src.data_ptr()), "synth")
if os.environ.get("UNAMB_POOL", os.environ.get("UNAMB_ARENA", "1")) != "0":   # outputs from the class pool
    pa, pb = ctx.alloc(8 * L), ctx.alloc(8 * L)
else:
    ta, tb = torch.empty(L, dtype=torch.int64, device=dev), torch.empty(L, dtype=torch.int64, device=dev)
    pa, pb = ta.data_ptr(), tb.data_ptr()
seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
res = cap.Result()
for K, J in ((21, 3), (31, 1)):
    for _ in range(2):
        rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, pa, pb, L, cap.MEM_DEVICE, C.byref(res))
        print(f"K {K} stride {J}: rc {rc} n_out {res.n_out}", flush=True)
torch.cuda.synchronize()
