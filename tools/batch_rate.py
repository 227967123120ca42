#!/usr/bin/env python3
"""Ragged batches through kmers_batch: N reads of a fixed or variable length from one resident pool,
CanonicalDNAMers{31} + fx_hash per read, everything in HBM.  Reports the time of the whole call
(layout kernels + recode + element kernel) and the rate in elements and source bases."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
K = int(os.environ.get("BATCH_K", "31"))   # BATCH_K=63: two-word kmers
NW = (2 * K + 63) // 64
res = cap.Result()
ARENA = "--pool" in sys.argv or "--arena" in sys.argv   # outputs from kmers_dev_alloc (the class pool) instead of torch
if "--passes" in sys.argv:   # force the tile length (1..8 passes of 1024 elements) instead of the per-call choice
    ctx.set_param(cap.PARAM_BATCH_PASSES, int(sys.argv[sys.argv.index("--passes") + 1]))
CASES = (("10 M reads x 150", 10_000_000, 150, 151, 4), ("10 M reads x 150", 10_000_000, 150, 151, 2),
                                   ("10 M reads x 150 (ASCII)", 10_000_000, 150, 151, 8),
                                   ("10 M reads x 150 in a FASTQ buffer", 10_000_000, 150, 151, -8),   # 150 of every 350 bytes
                                   ("4 M reads x 50..600", 4_000_000, 50, 600, 4), ("30 M reads x 36", 30_000_000, 36, 37, 2), ("100 k contigs x 2k..20k", 100_000, 2_000, 20_000, 4))
if "--quick" in sys.argv:
    CASES = CASES[:1]
for label, n_reads, lo, hi, src in CASES:
    rng = np.random.default_rng(1)
    lens = rng.integers(lo, hi, n_reads).astype(np.uint64)
    fastq = src == -8            # records lie 350 bytes apart: header, read, '+', qualities (here: more letters; never inspected)
    src = abs(src)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    n_pool = int(lens.sum())
    if fastq:
        starts = (np.arange(n_reads, dtype=np.uint64) * np.uint64(350) + np.uint64(40))
        n_pool = n_reads * 350
    if src == 8:
        pool = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", np.uint8), n_pool + 16)).to(dev)
        pool_ptr = pool.data_ptr()
    else:
        nw = (n_pool * src + 63) // 64
        pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 9, 0, nw, src, 0, pool.data_ptr()), "synth")
        pool_ptr = pool.data_ptr()
    spans_h = np.stack([starts, lens], axis=1).copy()
    spans_d = torch.from_numpy(spans_h.view(np.int64)).to(dev)
    total = int(np.maximum(lens.astype(np.int64) - K + 1, 0).sum())
    if ARENA:   # outputs from the device's class pool (two region classes), like bench.py's
        pk, ph = ctx.alloc(total * NW * 8), ctx.alloc(total * 8)
        out_k = out_h = None
    else:
        out_k = torch.empty(total * NW, dtype=torch.int64, device=dev)
        out_h = torch.empty(total, dtype=torch.int64, device=dev)
        pk, ph = out_k.data_ptr(), out_h.data_ptr()
    torch.cuda.synchronize()
    seq = cap.Seq(pool_ptr, n_pool, 0, 0, src, 0)
    for spans_ptr, flag, what in ((spans_h.ctypes.data, 0, "host spans"), (spans_d.data_ptr(), cap.SPANS_DEVICE, "resident spans")):
        best = 1e9
        for _ in range(4):
            t0 = time.perf_counter()
            rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans_ptr, n_reads, cap.BATCH_CANONICAL, K, 2, pk, ph,
                                     0, None, total, cap.MEM_DEVICE | flag, C.byref(res))
            best = min(best, time.perf_counter() - t0)
            assert rc == 0 and res.n_out == total, ctx.last_error()
        by = total * (8 * NW + 8) + int(lens.sum()) * src / 8
        print(f"src={src} {label:28s} {what:15s} {best * 1e3:8.3f} ms  {total / best / 1e9:7.1f} G elements/s  {int(lens.sum()) / best / 1e9:7.1f} Gbases/s  "
              f"{by / best / 1e9:7.0f} GB/s", flush=True)
    if ARENA:
        ctx.free(pk)
        ctx.free(ph)
    del pool, out_k, out_h, spans_d

# each_codon over the coding sequences of many genomes: SpacedDNAMers{3,3} per record (kmers_batch_spaced)
if "--quick" not in sys.argv:
    rng = np.random.default_rng(2)
    n_genes, src = 2_000_000, 4
    lens = (rng.integers(100, 900, n_genes) * 3).astype(np.uint64)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
    n_pool = int(lens.sum())
    nw = (n_pool * src + 63) // 64
    pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 9, 0, nw, src, 0, pool.data_ptr()), "synth")
    spans_d = torch.from_numpy(np.stack([starts, lens], axis=1).copy().view(np.int64)).to(dev)
    total = int((lens // 3).sum())
    out_k = torch.empty(total, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    seq = cap.Seq(pool.data_ptr(), n_pool, 0, 0, src, 0)
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans_d.data_ptr(), n_genes, 3, 3, 2, out_k.data_ptr(), None, total,
                                        cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0 and res.n_out == total, ctx.last_error()
    by = total * 8 + n_pool * src / 8
    print(f"src={src} {'2 M genes x 0.3..2.7 kb, each_codon':28s} {'resident spans':15s} {best * 1e3:8.3f} ms  {total / best / 1e9:7.1f} G elements/s  "
          f"{n_pool / best / 1e9:7.1f} Gbases/s  {by / best / 1e9:7.0f} GB/s", flush=True)
