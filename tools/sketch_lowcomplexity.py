#!/usr/bin/env python3
"""MinHash sketch rate on genome-like low-complexity input: 256 Mbase of random 2-bit sequence in which
a fraction of the positions sits in poly-A runs and short tandem repeats (their few, small hashes recur
millions of times).  Compares the device-resident path with the host-feedback path."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L = 256_000_000
rng = np.random.default_rng(1)
FRACS = (0.3,) if "--only30" in sys.argv else (0.0, 0.05, 0.3)
for frac in FRACS:
    codes = rng.integers(0, 4, L, dtype=np.uint8)
    pos = 0
    while frac and pos < L:          # runs of 500..5000 symbols covering about `frac` of the sequence
        gap = int(rng.integers(1000, 20000) * (1 - frac) / max(frac, 1e-9) / 10)
        pos += gap
        run = int(rng.integers(500, 5000))
        unit = [np.array(u, np.uint8) for u in ([0], [3], [0, 1], [0, 0, 2])][int(rng.integers(0, 4))]
        if pos >= L:
            break
        end = min(L, pos + run)
        codes[pos:end] = np.resize(unit, end - pos)
        pos = end
    low = float(np.mean(codes[:-1] == codes[1:]))
    words = np.zeros((L + 31) // 32, np.uint64)
    c = np.zeros(len(words) * 32, np.uint64)
    c[:L] = codes
    c = c.reshape(-1, 32)
    for j in range(32):
        words |= c[:, j] << np.uint64(2 * j)
    d = ctx.alloc(len(words) * 8 + 16)
    ctx.h2d(d, words)
    seq = cap.Seq(d, L, 0, 0, 2, 0)
    res = cap.Result()
    out = [np.zeros(1000, np.uint64), np.zeros(1000, np.uint64)]
    line = f"low-complexity fraction {frac:4.2f} (adjacent-equal rate {low:.3f}):"
    for host_only in (0, 1):
        ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, host_only)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ctx.check(ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, out[host_only].ctypes.data_as(C.c_void_p),
                                            cap.MEM_DEVICE, C.byref(res)), "minhash")
            best = min(best, time.perf_counter() - t0)
        line += f"  {'host-feedback' if host_only else 'device-resident'} {best * 1e3:8.3f} ms ({L / best / 1e9:6.1f} Gbases/s)"
    ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, 0)
    assert np.array_equal(out[0], out[1])
    print(line)
    ctx.free(d)
