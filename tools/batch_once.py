#!/usr/bin/env python3
"""bench.py's kmers_batch leg (8 M reads x 125 bases of one 1 Gbase LongDNA{4} pool, CanonicalDNAMers{31} + fx_hash per read),
three calls: the program rocprofv3 profiles for tools/r2_pmc_batch.sh.  With --passes P the tile is 1024 x P elements."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
res = cap.Result()
L = 1_000_000_000
nw = L // 16
buf = torch.empty(nw + 2, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw, 4, 0, buf.data_ptr()), "synth")
seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
n_reads, rl, Kb = 8_000_000, 125, 31
spans = torch.stack([torch.arange(n_reads, dtype=torch.int64, device=dev) * rl, torch.full((n_reads,), rl, dtype=torch.int64, device=dev)], dim=1).contiguous()
total = n_reads * (rl - Kb + 1)
a = torch.empty(total, dtype=torch.int64, device=dev)
b = torch.empty(total, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
passes = [int(p) for p in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
for p in passes:
    ctx.set_param(cap.PARAM_BATCH_PASSES, p)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.check(ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, Kb, 2, a.data_ptr(), b.data_ptr(), 0, None,
                                      total, cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res)), "batch")
        best = min(best, time.perf_counter() - t0)
    print(f"passes {p}: {best * 1e3:.3f} ms  {(16.0 * total + 0.5 * L) / best / 1e9 / 8000:.3f} of 8 TB/s", flush=True)
