#!/usr/bin/env python3
"""kmers_batch in the shape of bench.py's leg (8 M reads x 125 bases of a 4-bit pool, CanonicalDNAMers{31} + fx_hash per read,
everything resident, outputs from the library's allocator), three calls and nothing else: the program the profiler passes of
profiles/r05_batch.md / r06_batch.md run.
    python3 tools/batch_once.py [--reads N] [--len L] [--src 4|8] [--passes P] [--reps R] [--n-share 0.01 --skip] [--ragged]"""
import argparse
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=8_000_000)
ap.add_argument("--len", type=int, default=125)
ap.add_argument("--src", type=int, default=4)
ap.add_argument("--passes", type=int, default=0)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--dense", type=int, default=0, help="KMERS_PARAM_BATCH_DENSE: -1 = the general path only")
ap.add_argument("--n-share", type=float, default=0.0, help="share of the reads that hold one N")
ap.add_argument("--skip", action="store_true", help="KMERS_BATCH_SKIP")
ap.add_argument("--ragged", action="store_true", help="read lengths uniform in 50..250 instead of --len")
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
if args.passes:
    ctx.set_param(cap.PARAM_BATCH_PASSES, args.passes)
ctx.set_param(cap.PARAM_BATCH_DENSE, args.dense)
K, n_reads, rl, src = args.k, args.reads, args.len, args.src
rng = np.random.default_rng(1)
if args.ragged:
    lengths = rng.integers(50, 251, n_reads)
else:
    lengths = np.full(n_reads, rl, dtype=np.int64)
starts = np.concatenate([[0], np.cumsum(lengths[:-1])]).astype(np.int64)
n_pool = int(lengths.sum())
total = int((lengths - K + 1).sum())
pa, pb = ctx.alloc(8 * total), ctx.alloc(8 * total)
if src == 8:
    pool = torch.from_numpy(rng.choice(np.frombuffer(b"ACGT", np.uint8), n_pool + 16)).to(dev)
else:
    nw = (n_pool * src + 63) // 64
    pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 9, 0, nw, src, 0, pool.data_ptr()), "synth")
if args.n_share > 0:   # one N in a share of the reads
    hit = np.flatnonzero(rng.random(n_reads) < args.n_share)
    pos = starts[hit] + rng.integers(0, lengths[hit])
    if src == 8:
        pool[torch.from_numpy(pos).to(dev)] = 78
    else:
        per = 64 // src
        uw, inv = np.unique(pos // per, return_inverse=True)
        um = np.zeros(len(uw), dtype=np.uint64)
        np.bitwise_or.at(um, inv, np.uint64(0xF) << ((pos % per) * src).astype(np.uint64))
        idx = torch.from_numpy(uw.astype(np.int64)).to(dev)
        pool[idx] = torch.bitwise_or(pool[idx], torch.from_numpy(um.view(np.int64).copy()).to(dev))
spans = torch.from_numpy(np.stack([starts, lengths.astype(np.int64)], axis=1).copy()).to(dev)
torch.cuda.synchronize()
flags = cap.MEM_DEVICE | cap.SPANS_DEVICE | (cap.BATCH_SKIP if args.skip else 0)
seq = cap.Seq(pool.data_ptr(), n_pool, 0, 0, src, 0)
res = cap.Result()
ts = []
for _ in range(args.reps):
    t0 = time.perf_counter()
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, K, 2, pa, pb, 0, None, total,
                             flags, C.byref(res))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0 and res.n_out == total, ctx.last_error()
by = 16.0 * total + n_pool * src / 8
print(f"kmers_batch {n_reads} x {'50-250' if args.ragged else rl} src={src} N={args.n_share} skip={int(args.skip)} passes={args.passes} dense={args.dense}: {' '.join(f'{t:.3f}' for t in ts)} ms; best {by / min(ts) / 1e9:.2f} TB/s = {by / min(ts) / 1e9 / 8:.3f} of 8 TB/s")
