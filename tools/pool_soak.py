#!/usr/bin/env python3
"""The class pool under churn, THROUGH THE C ABI (the product's counterpart of tools/device_probes/vmm_churn.hip, which found the
GiB-boundary fault of profiles/r06_pool.md section 5): blocks of changing sizes and roles allocated, written at both ends by a
launch, checked, freed in any order; the cache switched off and on, the pool trimmed, a second context coming and going, and the
host's own allocations (torch: plain hipMalloc / hipFree) in between -- for a number of seconds, any wrong byte or fault fatal.

    python3 tools/pool_soak.py [--seconds 60] [--seed 1] [--max-gib 12]
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

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--max-gib", type=int, default=12)
args = ap.parse_args()

cap = km._capi
GiB, MiB = 1 << 30, 1 << 20
rng = np.random.default_rng(args.seed)
ctx = km.Context(0)
dev = torch.device("cuda", 0)
res = cap.Result()
K = 31


def h2d(c, ptr, arr):
    c.check(c.lib.kmers_memcpy_h2d(c.handle, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")


def d2h(c, ptr, n):
    back = np.zeros(n, dtype=np.uint64)
    c.check(c.lib.kmers_memcpy_d2h(c.handle, back.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), back.nbytes), "d2h")
    return back


# a source the launches read: 64 Mbase of 4-bit DNA
L = 64_000_000
nw = (L * 4 + 63) // 64
d_words = ctx.alloc(nw * 8 + 16)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 5, 0, nw, 4, 0, d_words), "synth")
seq = cap.Seq(d_words, L, 0, 0, 4, 0)
n = L - K + 1
ref = None

live = []        # (ctx, ptr, bytes, tag)
plain = []
t0 = time.perf_counter()
it = 0
other = None
counts = dict(allocs=0, frees=0, launches=0, trims=0, toggles=0, contexts=0, plain=0)
while time.perf_counter() - t0 < args.seconds:
    it += 1
    c = other if (other is not None and rng.integers(0, 3) == 0) else ctx
    # 1. one to three blocks: 130 MiB .. max GiB, a third of them lone outputs, sizes off the GiB grid
    for _ in range(int(rng.integers(1, 4))):
        size = int(rng.integers(130 * MiB, args.max_gib * GiB)) // 4096 * 4096
        if sum(b[2] for b in live) + size > 150 * GiB:
            break
        p = c.alloc(size, lone_output=bool(rng.integers(0, 3) == 0))
        tag = rng.integers(0, 1 << 63, 2, dtype=np.uint64)
        h2d(c, p, tag[:1])
        h2d(c, p + size - 8, tag[1:])
        live.append((c, p, size, tag))
        counts["allocs"] += 1
    # 2. a launch into a pair of blocks that are large enough (their heads are overwritten: the tags at the END stay)
    big = [b for b in live if b[2] >= n * 8 + 8]
    if len(big) >= 2:
        a, b = big[int(rng.integers(0, len(big)))], big[int(rng.integers(0, len(big)))]
        if a is not b:
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a[1], b[1], 0, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0, ctx.last_error()
            hk, hh = d2h(ctx, a[1], 4096), d2h(ctx, b[1], 4096)
            if ref is None:
                ref = (hk.copy(), hh.copy())
            assert np.array_equal(hk, ref[0]) and np.array_equal(hh, ref[1]), "a launch into blocks of the pool wrote something else"
            for blk in (a, b):
                head = d2h(blk[0], blk[1], 1)
                i = next(j for j, x in enumerate(live) if x is blk)
                live[i] = (blk[0], blk[1], blk[2], np.array([head[0], blk[3][1]], dtype=np.uint64))
            counts["launches"] += 1
    # 3. every block still shows its tags
    for (cc, p, size, tag) in live:
        assert d2h(cc, p, 1)[0] == tag[0] and d2h(cc, p + size - 8, 1)[0] == tag[1], f"block {hex(p)} of {size} bytes lost its data (iteration {it})"
    # 4. free some, in any order
    while live and (len(live) > 6 or rng.integers(0, 2)):
        j = int(rng.integers(0, len(live)))
        cc, p, size, tag = live.pop(j)
        (ctx if rng.integers(0, 2) else cc).free(p)
        counts["frees"] += 1
    # 5. the host's own memory: plain hipMalloc / hipFree through torch (no caching: emptied at once)
    for _ in range(int(rng.integers(0, 3))):
        if plain and rng.integers(0, 2):
            plain.pop(int(rng.integers(0, len(plain))))
            torch.cuda.empty_cache()
        x = torch.empty(int(rng.integers(1 * MiB, 3 * GiB)) // 8, dtype=torch.int64, device=dev)
        x.fill_(it)
        plain.append(x)
        counts["plain"] += 1
        if len(plain) > 8:
            plain.pop(0)
            torch.cuda.empty_cache()
    torch.cuda.synchronize()
    # 6. now and then: the cache off / on, a trim, a second context that comes or goes
    r = int(rng.integers(0, 12))
    if r == 0:
        ctx.set_param(cap.PARAM_POOL_CACHE, int(rng.integers(0, 2)))
        counts["toggles"] += 1
    elif r == 1:
        ctx.pool_trim()
        counts["trims"] += 1
    elif r == 2:
        if other is None:
            other = km.Context(0)
            counts["contexts"] += 1
        elif not any(b[0] is other for b in live):
            other.close()
            other = None
    if it % 20 == 0:
        st = ctx.pool_stats()
        print(f"iteration {it} at {time.perf_counter() - t0:.0f} s: {counts}, held {st['held'] / GiB:.1f} GiB, in use {st['in_use'] / GiB:.1f}, "
              f"cached {st['cached'] / GiB:.1f}, handles created {st['chunks_created']} / returned {st['chunks_returned']}", flush=True)

for (cc, p, size, tag) in live:
    ctx.free(p)
st = ctx.pool_stats()
print(f"ok: {it} iterations, {counts}; handles created {st['chunks_created']}, returned {st['chunks_returned']}, cache hits {st['cache_hits']}, "
      f"blocks assembled {st['cache_misses']}, evictions {st['evictions']}")
ctx.free(d_words)
if other is not None:
    other.close()
ctx.close()
