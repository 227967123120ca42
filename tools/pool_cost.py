#!/usr/bin/env python3
"""What a block of the class pool costs ("price the class pool on the collect path"): kmers_dev_alloc and kmers_dev_free
in microseconds of host time by size -- the FIRST block of a size (the pool creates handles, measures their class, maps, checks every
handle), the same size again after a free (the pool's cache: no call into the driver), and with the cache switched off
(KMERS_PARAM_POOL_CACHE = 0: the round-5 shape of a free -- wait, unmap, flush -- for comparison).  One fresh process.

    python3 tools/pool_cost.py [--sizes-gib 0.25,1,8,16,80] [--reps 20]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import kmers_jl_amd as km

ap = argparse.ArgumentParser()
ap.add_argument("--sizes-gib", default="0.25,1,8,16,80")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
lib, h = ctx.lib, ctx.handle
GiB = 1 << 30


def alloc(nbytes):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = lib.kmers_dev_alloc(h, nbytes, C.byref(p))
    dt = (time.perf_counter() - t0) * 1e6
    assert rc == 0, ctx.last_error()
    return p.value, dt


def free(p):
    t0 = time.perf_counter()
    rc = lib.kmers_dev_free(h, C.c_void_p(p))
    dt = (time.perf_counter() - t0) * 1e6
    assert rc == 0, ctx.last_error()
    return dt


# the empty ctypes call: what the numbers below contain that is not the library's
t0 = time.perf_counter()
for _ in range(1000):
    lib.kmers_abi_version()
call_us = (time.perf_counter() - t0) * 1e6 / 1000
print(f"# an empty call through ctypes: {call_us:.2f} us (contained in every figure below)")
print("| size | first alloc (pair: a, then b beside it) us | first free us | cached alloc us (median of pairs) | free us (median) | cache off: alloc us | cache off: free us | pool after |")
print("|---|---|---|---|---|---|---|---|")
for g in [float(x) for x in args.sizes_gib.split(",")]:
    n = int(g * GiB)
    st0 = ctx.pool_stats()
    a, ta = alloc(n)
    b, tb = alloc(n)
    fa, fb = free(a), free(b)
    al, fr = [], []
    for _ in range(args.reps):
        a, t1 = alloc(n)
        b, t2 = alloc(n)
        al += [t1, t2]
        fr += [free(a), free(b)]
    st1 = ctx.pool_stats()
    ctx.set_param(cap.PARAM_POOL_CACHE, 0)
    al0, fr0 = [], []
    for _ in range(max(3, args.reps // 4)):
        a, t1 = alloc(n)
        b, t2 = alloc(n)
        al0 += [t1, t2]
        fr0 += [free(a), free(b)]
    ctx.set_param(cap.PARAM_POOL_CACHE, 1)
    st2 = ctx.pool_stats()
    print(f"| {g:g} GiB | {ta:.0f} + {tb:.0f} | {fa:.1f} + {fb:.1f} | {np.median(al):.1f} | {np.median(fr):.1f} | {np.median(al0):.0f} | {np.median(fr0):.0f} | "
          f"held {st2['held'] / GiB:.0f} GiB, cached {st2['cached'] / GiB:.0f}, hits {st1['cache_hits'] - st0['cache_hits']}, "
          f"handles created {st2['chunks_created'] - st0['chunks_created']}, returned {st2['chunks_returned'] - st0['chunks_returned']} |", flush=True)
    ctx.pool_trim()
ctx.close()
