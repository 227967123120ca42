#!/usr/bin/env python3
"""kmers_batch in the shape of bench.py's leg (8 M reads x 125 bases of a 4-bit pool, CanonicalDNAMers{31} + fx_hash per read,
everything resident, outputs from the library's allocator), three calls and nothing else: the program the profiler passes of
profiles/r05_batch.md run.    python3 tools/batch_once.py [--reads N] [--len L] [--src 4] [--passes P] [--reps R]"""
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
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
if args.passes:
    ctx.set_param(cap.PARAM_BATCH_PASSES, args.passes)
ctx.set_param(cap.PARAM_BATCH_DENSE, args.dense)
K, n_reads, rl, src = args.k, args.reads, args.len, args.src
n_pool = n_reads * rl
total = n_reads * (rl - K + 1)
pa, pb = ctx.alloc(8 * total), ctx.alloc(8 * total)
if src == 8:
    pool = torch.from_numpy(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", np.uint8), n_pool + 16)).to(dev)
else:
    nw = (n_pool * src + 63) // 64
    pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 9, 0, nw, src, 0, pool.data_ptr()), "synth")
spans = torch.stack([torch.arange(n_reads, dtype=torch.int64, device=dev) * rl, torch.full((n_reads,), rl, dtype=torch.int64, device=dev)], dim=1).contiguous()
torch.cuda.synchronize()
seq = cap.Seq(pool.data_ptr(), n_pool, 0, 0, src, 0)
res = cap.Result()
ts = []
for _ in range(args.reps):
    t0 = time.perf_counter()
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, K, 2, pa, pb, 0, None, total,
                             cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res))
    ts.append((time.perf_counter() - t0) * 1e3)
    assert rc == 0 and res.n_out == total, ctx.last_error()
by = 16.0 * total + n_pool * src / 8
print(f"kmers_batch {n_reads} x {rl} src={src} passes={args.passes} dense={args.dense}: {' '.join(f'{t:.3f}' for t in ts)} ms; best {by / min(ts) / 1e9:.2f} TB/s = {by / min(ts) / 1e9 / 8:.3f} of 8 TB/s")
