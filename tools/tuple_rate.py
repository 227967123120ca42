#!/usr/bin/env python3
"""C2 with the two output layouts: separate kmers[] / hashes[] arrays (default) vs one array of
Tuple{Kmer,UInt64} elements (KMERS_OUT_TUPLES), interleaved A/B on one box."""
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
L, K, bits = 1_000_000_000, 31, 4
nw = (L * bits + 63) // 64
n = L - K + 1
res = cap.Result()
with torch.cuda.stream(stream):
    buf = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 3, 0, nw, bits, 0, buf.data_ptr()), "synth")
    a = torch.empty(n, dtype=torch.int64, device=dev)
    b = torch.empty(n, dtype=torch.int64, device=dev)
    t = torch.empty(2 * n, dtype=torch.int64, device=dev)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    tiles = [int(x) for x in sys.argv[1:]] or [0]

    def run(tuples, tile):
        ctx.set_param(cap.PARAM_TILE_KMERS, tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        if tuples:
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, t.data_ptr(), None, 0, cap.MEM_DEVICE | cap.OUT_TUPLES, C.byref(res))
        else:
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, cap.MEM_DEVICE, C.byref(res))
        e1.record(stream)
        torch.cuda.synchronize()
        assert rc == 0
        return e0.elapsed_time(e1)
    for tile in tiles:
        ts = {0: [], 1: []}
        for rep in range(9):
            for tup in (0, 1):
                ts[tup].append(run(tup, tile))
        for tup in (0, 1):
            ms = float(np.median(ts[tup][2:]))
            print(f"tile={tile or 'default':>7} {'tuples (one array)' if tup else 'two arrays':20s} {ms:7.3f} ms  {16.5 * n / ms / 1e6:7.0f} GB/s  {16.5 * n / ms / 1e6 / 80:5.1f} % of 8 TB/s")
    # tuples must equal the two arrays interleaved
    ok = bool(torch.equal(t.view(-1, 2)[:, 0], a)) and bool(torch.equal(t.view(-1, 2)[:, 1], b))
    print("tuple layout == interleaved arrays:", ok)
