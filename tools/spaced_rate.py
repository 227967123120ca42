#!/usr/bin/env python3
"""kmers_spaced over 1 Gbase LongDNA{4} at strides on both sides of what a tile of the stream kernel stages (J * bits <= 64):
SpacedKmers{K,K} (non-overlapping kmers) below and above K = 32.  Device-resident, HIP events, best of 4."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmers_jl_amd as km  # noqa: E402

cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
res = cap.Result()
with torch.cuda.stream(stream):
    nw = L * 4 // 64 + 1
    buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, 4, 0, buf.data_ptr()), "synth")
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
    for K, J in ((21, 3), (31, 31), (32, 32), (33, 33), (40, 40), (64, 64), (100, 100), (128, 128), (31, 100), (31, 1000), (150, 150), (150, 3)):
        N = (2 * K + 63) // 64
        n = (L - K) // J + 1
        a = torch.empty(n * N, dtype=torch.int64, device=dev)
        for no_tiles in (0, 1):
            ctx.set_param(cap.PARAM_WIDE_NO_TILES, no_tiles)
            best = 1e9
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, a.data_ptr(), cap.MEM_DEVICE | cap.ASYNC, C.byref(res))
                e1.record(stream)
                torch.cuda.synchronize()
                assert rc == 0 and ctx.sync()[0] == 0, ctx.last_error()
                best = min(best, e0.elapsed_time(e1))
            gb = (n * 8 * N + L / 2) / 1e9
            print(f"K {K:4d} J {J:5d} N {N}: no_tiles={no_tiles} {best:8.3f} ms {gb / best * 1e3:8.1f} GB/s (whole source counted)", flush=True)
        del a
