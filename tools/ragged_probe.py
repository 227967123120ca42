#!/usr/bin/env python3
"""Phase times inside ragged_kernel (diagnostic build: -DKMERS_RG_PROBE, selected with KMERS_HIP_LIB).
    hipcc ... -DKMERS_RG_PROBE -o /path/libkmers_probe.so ; KMERS_HIP_LIB=/path/libkmers_probe.so python tools/ragged_probe.py
Lane 0 of every wavefront of one workgroup in 128 adds (clock - clock at entry) at six points; the averages,
in core-clock cycles, say where a wavefront's life goes."""
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
K = 31
res = cap.Result()
probe = (C.c_ulonglong * 16)()
for label, n_reads, lo, hi, src in (("10 M reads x 150", 10_000_000, 150, 151, 2), ("10 M reads x 150", 10_000_000, 150, 151, 4),
                                   ("100 k contigs", 100_000, 2_000, 20_000, 4)):
    rng = np.random.default_rng(1)
    lens = rng.integers(lo, hi, n_reads).astype(np.uint64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    n_pool = int(lens.sum())
    nw = (n_pool * src + 63) // 64
    pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 9, 0, nw, src, 0, pool.data_ptr()), "synth")
    spans_d = torch.from_numpy(np.stack([starts, lens], axis=1).copy().view(np.int64)).to(dev)
    total = int(np.maximum(lens.astype(np.int64) - K + 1, 0).sum())
    out_k = torch.empty(total, dtype=torch.int64, device=dev)
    out_h = torch.empty(total, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    seq = cap.Seq(pool.data_ptr(), n_pool, 0, 0, src, 0)
    for it in range(3):
        ctx.lib.kmers_debug_rg_probe(probe, 1)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans_d.data_ptr(), n_reads, cap.BATCH_CANONICAL, K, 2, out_k.data_ptr(),
                                 out_h.data_ptr(), 0, None, total, cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res))
        assert rc == 0
        ctx.lib.kmers_debug_rg_probe(probe, 0)
    n = probe[0]
    names = ["", "loads issued", "loads landed (barrier 1)", "owners scanned (barriers 2-4)", "elements computed", "stores issued", "stores drained"]
    print(f"src={src} {label}: {n} wavefronts sampled; cycles since entry:")
    prev = 0
    for i in range(1, 7):
        t = probe[i] / max(n, 1)
        print(f"   {names[i]:32s} {t:9.0f}  (+{t - prev:.0f})")
        prev = t
    del pool, out_k, out_h, spans_d
