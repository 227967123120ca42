#!/usr/bin/env python3
"""Stress of the single-pass UnambiguousKmers kernel (inter-workgroup look-back: rare-event bugs do not show in a handful of
runs): random lengths, K, stride lattices, ambiguity patterns (i.i.d., long N blocks, nothing / everything dropped), tile sizes
and grid caps, device outputs compared element by element with the oracle.    python tools/stress_unamb.py [first_seed] [n]"""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
from oracle import pyoracle
cap = km._capi
orc = pyoracle.get()
ctx = km.Context(0)
dev = torch.device("cuda", 0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t_start = time.time()
worst = 0.0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    L = int(rng.choice([rng.integers(1, 3000), rng.integers(3000, 300_000), rng.integers(300_000, 40_000_000)]))
    K = int(rng.choice([1, 3, 21, 31, 33, 64, 65, 128]))
    stride = int(rng.choice([1, 1, 1, 3, 7, 100]))
    nw = (L * 4 + 63) // 64
    mode = seed % 5
    if mode == 0:
        words = orc.synth_words(seed, 0, nw + 1, 4, 2621)                      # p(N) = 0.04
    elif mode == 1:
        words = orc.synth_words(seed, 0, nw + 1, 4, int(rng.integers(0, 3)))    # (almost) nothing dropped
    elif mode == 2:
        words = orc.synth_words(seed, 0, nw + 1, 4, 40000)                     # most windows dropped
    else:
        words = orc.synth_words(seed, 0, nw + 1, 4, 0).copy()                  # clean sequence with N blocks of random length
        w = words
        for _ in range(int(rng.integers(1, 40))):
            a = int(rng.integers(0, max(nw, 1)))
            b = min(nw, a + int(rng.integers(1, max(2, nw // 10))))
            w[a:b] = np.uint64(0xFFFFFFFFFFFFFFFF) if mode == 3 else np.uint64(0)   # N ... or gaps
    ek, es, _ = orc.unambiguous(words, L, 4, K)
    keep = (es - 1) % stride == 0
    ek, es = ek[keep], es[keep]
    n = len(ek)
    N = (2 * K + 63) // 64
    d_src = torch.from_numpy(words.view(np.int64)).to(dev)
    room = n + 64
    dk = torch.full((room * N,), -1, dtype=torch.int64, device=dev)
    ds = torch.full((room,), -1, dtype=torch.int64, device=dev)
    tile = int(rng.choice([0, 0, 1024, 4096, 8192, 32768]))
    grid = int(rng.choice([0, 0, 1, 3, 64, 700]))
    ctx.set_param(cap.PARAM_TILE_KMERS, tile)
    ctx.set_param(cap.PARAM_MAX_GRID, grid)
    seq = cap.Seq(d_src.data_ptr(), L, 0, 0, 4, 0)
    res = cap.Result()
    torch.cuda.synchronize()
    t0 = time.time()
    rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, stride, dk.data_ptr(), ds.data_ptr(), room, cap.MEM_DEVICE, C.byref(res))
    dt = time.time() - t0
    worst = max(worst, dt)
    assert rc == 0 and res.n_out == n, (seed, rc, res.n_out, n, ctx.last_error())
    gk = dk.cpu().numpy().view(np.uint64).reshape(room, N)
    gs = ds.cpu().numpy()
    assert np.array_equal(gk[:n], ek) and np.array_equal(gs[:n], es), (seed, L, K, stride, mode, tile, grid)
    assert np.all(gs[n:] == -1), (seed, "wrote beyond the count")
    if (seed - first) % 20 == 19:
        print(f"seed {seed}: ok ({time.time() - t_start:.0f} s, slowest call {worst * 1e3:.1f} ms)", flush=True)
print(f"{count} cases from seed {first}: all equal to the oracle; slowest call {worst * 1e3:.1f} ms")
