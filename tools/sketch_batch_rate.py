#!/usr/bin/env python3
"""kmers_minhash_batch: one sketch (s = 1000, CanonicalDNAMers{16}) per record of a resident pool, against a loop of
kmers_minhash calls over the same records."""
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
K, s = 16, 1000
if "--s" in sys.argv:        # sketch size (1000: docs/src/minhash.md:34)
    s = int(sys.argv[sys.argv.index("--s") + 1])
res = cap.Result()
if "--lds" in sys.argv:      # force the candidate buffer (2048 / 4096 / 8192 values per workgroup)
    ctx.set_param(cap.PARAM_SKETCH_BATCH_LDS, int(sys.argv[sys.argv.index("--lds") + 1]))
for label, n_rec, lo, hi in (("100 k records x 5-15 kbases", 100_000, 5_000, 15_000), ("10 k genomes x 50-150 kbases", 10_000, 50_000, 150_000),
                             ("1 M reads x 1 kbase", 1_000_000, 1_000, 1_001)):
    rng = np.random.default_rng(2)
    lens = rng.integers(lo, hi, n_rec).astype(np.uint64)
    lens16 = (lens + 15) // 16 * 16                      # every record starts on a word boundary of the 4-bit pool
    starts = np.concatenate([[0], np.cumsum(lens16)[:-1]]).astype(np.uint64)
    n_pool = int(lens16.sum())
    nw = n_pool // 16
    pool = torch.empty(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 11, 0, nw, 4, 0, pool.data_ptr()), "synth")
    spans = torch.from_numpy(np.stack([starts, lens], axis=1).copy().view(np.int64)).to(dev)
    out = torch.empty(n_rec * s, dtype=torch.int64, device=dev)
    cnt = torch.empty(n_rec, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    seq = cap.Seq(pool.data_ptr(), n_pool, 0, 0, 4, 0)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_rec, K, 2, 0, s, out.data_ptr(), cnt.data_ptr(),
                                         cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0, ctx.last_error()
    # the same sketches one call at a time (a sample of records, extrapolated)
    sample = min(n_rec, 2000)
    one = np.zeros(s, np.uint64)
    t0 = time.perf_counter()
    for i in range(sample):
        v = cap.Seq(pool.data_ptr(), int(lens[i]), int(starts[i]), 0, 4, 0)
        ctx.lib.kmers_minhash(ctx.handle, C.byref(v), K, 2, 0, s, one.ctypes.data, cap.MEM_DEVICE, C.byref(res))
    loop = (time.perf_counter() - t0) / sample * n_rec
    chk = out.view(n_rec, s)[sample - 1, :int(res.n_out)].cpu().numpy().view(np.uint64)
    assert np.array_equal(chk, one[:int(res.n_out)])
    print(f"{label:30s} batch {best * 1e3:9.2f} ms ({int(lens.sum()) / best / 1e9:6.1f} Gbases/s, {n_rec / best / 1e3:8.1f} k sketches/s)   "
          f"per-record calls {loop * 1e3:10.1f} ms (extrapolated from {sample})   x{loop / best:6.0f}", flush=True)
    del pool, out, cnt, spans
