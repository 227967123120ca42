#!/usr/bin/env python3
"""C2 (CanonicalDNAMers{31} + fx_hash over LongDNA{4}) from 10^5 to 10^10 symbols: where the fixed cost of a call,
the PCIe hops and the placement of the output arrays bite, and whether `KmersHIP.MIN_BASES[] = 100 000` is where the device starts
to win (src/iterators/FwKmers.jl:14-22: tiny inputs must not regress; docs/src/kmers.md:129-135: the reference's ~1 ns per symbol).

Per length, one fresh pair of output arrays from kmers_dev_alloc (what a host gets: the class pool from 128 MiB on, hipMalloc below):
  kernel        KMERS_MEM_DEVICE | KMERS_ASYNC launches back to back, HIP events around each (median): the launch alone
  sync call     KMERS_MEM_DEVICE, synchronous: wall clock per call (launch + the wait + the 16-byte status copy)
  host call     KMERS_MEM_HOST (numpy arrays: pageable memory): wall clock per call, words up + kmers and hashes down
  cpu port      the oracle (oracle/kmers_oracle.c, -O3), one thread, same input

    python3 tools/size_sweep.py [--max-host 256000000] [--max-cpu 30000000] [--plain]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import kmers_jl_amd as km
from oracle import pyoracle

ap = argparse.ArgumentParser()
ap.add_argument("--lengths", default="10000,30000,100000,300000,1000000,3000000,10000000,32000000,64000000,100000000,320000000,1000000000,10000000000")
ap.add_argument("--max-host", type=int, default=256_000_000)
ap.add_argument("--max-cpu", type=int, default=30_000_000)
ap.add_argument("--plain", action="store_true", help="KMERS_PARAM_POOL = 0: every array a plain hipMalloc (A/B)")
args = ap.parse_args()

cap = km._capi
ctx = km.Context(0)
if args.plain:
    ctx.set_param(cap.PARAM_POOL, 0)
orc = pyoracle.get()
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
K = 31
ASYNC = cap.MEM_DEVICE | cap.ASYNC
print(f"# C2 by length; outputs from kmers_dev_alloc ({'plain hipMalloc (KMERS_PARAM_POOL = 0)' if args.plain else 'class pool from 128 MiB on'}); fractions of 8 TB/s over 16.5 B per kmer")
print("| symbols | arrays | kernel us | kernel frac | sync call us | sync Gbases/s | host call ms | host Gbases/s | cpu port ms (1 thread) | cpu Gbases/s | device wins from host memory |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for L in [int(x) for x in args.lengths.split(",")]:
    n, nw = L - K + 1, (L * 4 + 63) // 64
    try:
        d_w = ctx.alloc(nw * 8 + 16)
        d_k, d_h = ctx.alloc(n * 8), ctx.alloc(n * 8)
    except km.KmersError as e:
        print(f"| {L} | {e} |")
        break
    lay = lambda p: "".join("ABC?"[c] for c in ctx.pool_layout(p)[1])
    where = f"{lay(d_k) or 'hipMalloc'} / {lay(d_h) or 'hipMalloc'}"
    if len(where) > 40:
        where = f"{len(lay(d_k))} + {len(lay(d_h))} handles"
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 99, 0, nw, 4, 0, d_w), "synth")
    seq = cap.Seq(d_w, L, 0, 0, 4, 0)
    launch = lambda flags: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, d_k, d_h, 0, flags, C.byref(res))
    reps = 200 if L <= 10_000_000 else 30 if L <= 1_000_000_000 else 5
    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:                   # a busy device (an idle one runs its next milliseconds slower)
            for _ in range(8):
                assert launch(ASYNC) == 0
            torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            assert launch(ASYNC) == 0
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
    kern_us = float(np.median([a.elapsed_time(b) for a, b in evs])) * 1e3
    assert ctx.sync()[0] == 0
    t0 = time.perf_counter()
    for _ in range(reps):
        assert launch(cap.MEM_DEVICE) == 0
    sync_us = (time.perf_counter() - t0) / reps * 1e6
    host_ms = cpu_ms = None
    if L <= args.max_host:
        words = np.zeros(nw + 2, np.uint64)
        ctx.d2h(words, d_w)
        hk, hh = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
        hseq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
        call = lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(hseq), K, 2, hk.ctypes.data_as(C.c_void_p), hh.ctypes.data_as(C.c_void_p), 0,
                                               cap.MEM_HOST, C.byref(res))
        assert call() == 0
        hreps = 20 if L <= 10_000_000 else 3
        host_ms = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(hreps):
                assert call() == 0
            host_ms = min(host_ms, (time.perf_counter() - t0) / hreps * 1e3)
        if L <= args.max_cpu:
            ok, oh = np.ones((n, 1), np.uint64), np.ones(n, np.uint64)   # (touched: the timing must not hold the page faults of fresh arrays)
            cpu_ms = 1e30
            for _ in range(5 if L <= 3_000_000 else 2):
                t0 = time.perf_counter()
                ek, eh, _ = orc.canonical(words, L, 4, 2, K, out=ok, out_h=oh)
                cpu_ms = min(cpu_ms, (time.perf_counter() - t0) * 1e3)
            assert np.array_equal(ek[:, 0], hk) and np.array_equal(eh, hh), "the host call and the oracle disagree"
    frac = 16.5 * n / (kern_us * 1e-6) / 8e12
    cell = lambda v, f: f.format(v) if v is not None else ""
    wins = "" if host_ms is None or cpu_ms is None else ("yes" if host_ms < cpu_ms else "no")
    print(f"| {L} | {where} | {kern_us:.1f} | {frac:.3f} | {sync_us:.1f} | {L / sync_us / 1e3:.1f} | {cell(host_ms, '{:.3f}')} | "
          f"{cell(None if host_ms is None else L / host_ms / 1e6, '{:.2f}')} | {cell(cpu_ms, '{:.2f}')} | {cell(None if cpu_ms is None else L / cpu_ms / 1e6, '{:.2f}')} | {wins} |",
          flush=True)
    for p in (d_k, d_h, d_w):
        ctx.free(p)
    ctx.pool_trim()
ctx.close()
