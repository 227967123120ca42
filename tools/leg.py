#!/usr/bin/env python3
"""One leg of the hot path in ONE fresh process: the program every timing sweep and every `rocprofv3` pass of round 3 runs.

    python3 tools/leg.py --leg c2|f1|c3|c4|c4t|c5|u31|u21|xor|minhash|comp8 [--alloc pool|plain|carve:GB|prefree:GB] [--reps N]
                         [--busy-ms MS] [--tile T] [--once]

legs (1 Gbase LongDNA{4} unless stated; algorithmic bytes per SURVEY.md section 8d):
  c2   CanonicalDNAMers{31} + fx_hash, two arrays (16.5 B/kmer)          c3   CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2} (8.25)
  c4   FwDNAMers{63} + reverse complements, two arrays (32.5)            c4t  the same as one array of Tuple{Kmer,Kmer}
  c5   SpacedDNAMers{21,3} strict (0.5 B/base + 8 B/kmer)                u31 / u21  UnambiguousDNAMers{31} / the stride-3 lattice
  xor / minhash / comp8   the fused consumers (no HBM roofline)              of K = 21 at p(N) = 0.04 (0.5 B/base + 16 B/kept)
alloc (where the output arrays come from -- profiles/r02_tuning.md section 7, profiles/r03_alloc.md):
  plain       torch allocations (hipMalloc), first allocations of the process
  carve:GB    carved out of one torch allocation of GB gigabytes
  prefree:GB  GB gigabytes allocated and released first, then plain
  pool        kmers_dev_alloc: the device's class pool (the product's default)
--once: two launches and nothing else (the form the PMC passes profile).
Prints one line: leg, alloc, median ms, fraction of 8 TB/s (materialising legs).
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
ap.add_argument("--leg", default="c2")
ap.add_argument("--alloc", default="pool")
ap.add_argument("--reps", type=int, default=9)
ap.add_argument("--busy-ms", type=float, default=300.0)
ap.add_argument("--tile", type=int, default=0)
ap.add_argument("--max-grid", type=int, default=0)
ap.add_argument("--subtiles", type=int, default=0)
ap.add_argument("--threads", type=int, default=0, help="threads per workgroup of the tile kernels (KMERS_PARAM_BLOCK_THREADS)")
ap.add_argument("--split", action="store_true", help="two write windows per array (KMERS_PARAM_SPLIT_ORDER)")
ap.add_argument("--once", action="store_true")
ap.add_argument("--no-role", action="store_true", help="pool mode: a single output array is allocated like any other block (not by KMERS_ALLOC_LONE_OUTPUT)")
ap.add_argument("--shifts", default="", help="carve mode: comma-separated SA:SB byte shifts of the two output bases inside the block; one timing per pair, same process")
ap.add_argument("--bases", type=int, default=1_000_000_000)
args = ap.parse_args()

cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
GOLDEN = 0x9E3779B97F4A7C15
ASYNC = cap.MEM_DEVICE | cap.ASYNC
if args.tile:
    ctx.set_param(cap.PARAM_TILE_KMERS, args.tile)
if args.max_grid:
    ctx.set_param(cap.PARAM_MAX_GRID, args.max_grid)
if args.subtiles:
    ctx.set_param(cap.PARAM_SUBTILES, args.subtiles)
if args.threads:
    ctx.set_param(cap.PARAM_BLOCK_THREADS, args.threads)
if args.split:
    ctx.set_param(cap.PARAM_SPLIT_ORDER, 1)

leg = args.leg
if leg == "n1":  # the north star: the C2 launch over 10 Gbase on one GPU
    leg, args.bases = "c2", 10_000_000_000
L = 1_250_000_000 if leg == "c3" else args.bases
bits = 2 if leg == "c3" else 8 if leg == "f1" else 4
amb = 2621 if leg in ("u31", "u21") else 0
seed = {"c2": 2, "c3": 3, "c4": 4, "c4t": 4, "c5": 5, "u31": 5, "u21": 5}.get(leg, 5)
K = {"f127": 127, "c63": 63, "c127h": 127, "c63h": 63, "f3": 31, "c2": 31, "f1": 31, "c3": 31, "c4": 63, "c4t": 63, "c5": 21, "u31": 31, "u21": 21, "xor": 31, "minhash": 16, "minhash31": 31, "comp8": 8, "comp4": 4, "comp6": 6, "f4h": 31, "f4r": 31}[leg]
J = 3 if leg in ("c5", "u21") else 1
n = (L - K) // J + 1
words_a = {"f4h": L, "f4r": L, "f127": 4 * n, "c63": 2 * n, "c127h": 4 * n, "c63h": 2 * n, "f3": 2 * n, "c2": n, "f1": n, "c3": n, "c4": 2 * n, "c4t": 4 * n, "c5": n, "u31": n, "u21": n}.get(leg, 1 << 16)
words_b = {"f4h": L, "f4r": L, "f127": 4 * n, "c127h": n, "c63h": n, "c2": n, "f1": n, "c4": 2 * n, "u31": n, "u21": n}.get(leg, 0)

mode, _, size = args.alloc.partition(":")
size = int(size or 0)
keep = []
with torch.cuda.stream(stream):
    nw = (L * bits + 63) // 64
    src = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
    if leg == "f1":  # ASCII text: "ACGT" drawn uniformly
        idx = torch.randint(0, 4, (L,), dtype=torch.uint8, device=dev)
        text = src.view(torch.uint8)
        text.fill_(65)
        for code, add in ((1, 2), (2, 6), (3, 19)):
            text[:L] += idx.eq(code).to(torch.uint8) * add
        del idx
    else:
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, GOLDEN ^ seed, 0, nw, bits, amb, src.data_ptr()), "synth")
    if mode == "prefree":
        tmp = torch.empty(size * (1 << 30) // 8, dtype=torch.int64, device=dev)
        del tmp
        torch.cuda.empty_cache()
    if mode == "carve":
        block = torch.empty(size * (1 << 30) // 8, dtype=torch.int64, device=dev)
        keep.append(block)
        pa = block.data_ptr()
        pb = pa + ((8 * words_a + (1 << 21) - 1) >> 21 << 21)
    elif mode == "pool":
        t0 = time.perf_counter()
        pa = ctx.alloc(8 * words_a, lone_output=(words_b == 0 and not args.no_role))  # the only output of its launch: by role
        pb = ctx.alloc(8 * max(words_b, 1))
        info = ctx.pool_info()
        lay = lambda p: "".join("ABC?"[c] for c in ctx.pool_layout(p)[1])
        la, lb = lay(pa), lay(pb)
        short = lambda t: t if len(t) <= 48 else t[:40] + "..." + t[-5:]
        print(f"pool: {time.perf_counter() - t0:.2f} s; held {info['held'] / 2**30:.1f} GiB, in use {info['in_use'] / 2**30:.1f}, classes {info['n_classes']} "
              f"{[round(b / 2**30, 1) for b in info['class_bytes']]} GiB, probes two-class {info['two_class_gbps']:.0f} one-class {info['one_class_gbps']:.0f} GB/s; "
              f"a {short(la)} b {short(lb)}", flush=True)
    else:
        ta = torch.empty(words_a, dtype=torch.int64, device=dev)
        tb = torch.empty(max(words_b, 1), dtype=torch.int64, device=dev)
        keep += [ta, tb]
        pa, pb = ta.data_ptr(), tb.data_ptr()
torch.cuda.synchronize()
seq = cap.Seq(src.data_ptr(), L, 0, 0, bits, 0)
val = C.c_uint64()
sk = np.zeros(1000, dtype=np.uint64)
m_kept = 0
if leg in ("u31", "u21"):
    ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
    m_kept = int(res.n_out)

calls = {
    "c2": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, ASYNC, C.byref(res)),
    "f1": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, ASYNC, C.byref(res)),   # the headline launch from ASCII text
    "c3": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, None, 0, ASYNC, C.byref(res)),
    "f127": lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, pa, pb, ASYNC, C.byref(res)),   # four-word kmers + reverse complements
    "c63": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, None, 0, ASYNC, C.byref(res)),           # two-word canonical kmers, one array
    "c127h": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, ASYNC, C.byref(res)),            # four-word canonical kmers + hashes
    "c63h": lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pa, pb, 0, ASYNC, C.byref(res)),   # two-word canonical kmers + hashes
    "f3": lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 4, pa, None, ASYNC, C.byref(res)),   # 4-bit kmers, two words, one array
    "c4": lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, pa, pb, ASYNC, C.byref(res)),
    "c4t": lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, pa, None, ASYNC | cap.OUT_TUPLES, C.byref(res)),
    "c5": lambda: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, pa, ASYNC, C.byref(res)),
    "u31": lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, pa, pb, m_kept, ASYNC, C.byref(res)),
    "u21": lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, pa, pb, m_kept, ASYNC, C.byref(res)),
    "xor": lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)),
    "minhash": lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)),
    "comp8": lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, pa, cap.MEM_DEVICE, C.byref(res)),
    "f4h": lambda: ctx.lib.kmers_fx_hash(ctx.handle, pa, 1, L, 0, pb, ASYNC),                        # fx_hash over an array of 1 G one-word kmers
    "f4r": lambda: ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, pa, K, 2, L, pb, ASYNC),       # reverse_complement over the same
    "comp4": lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, pa, cap.MEM_DEVICE, C.byref(res)),
    "comp6": lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, pa, cap.MEM_DEVICE, C.byref(res)),
    "minhash31": lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)),
}
alg = {"f4h": 16.0 * L, "f4r": 16.0 * L, "f127": 64.5 * n, "c63": 16.5 * n, "c127h": 40.5 * n, "c63h": 24.5 * n, "f3": 16.5 * n, "c2": 16.5 * n, "f1": 17.0 * n, "c3": 8.25 * n, "c4": 32.5 * n, "c4t": 32.5 * n, "c5": 0.5 * L + 8.0 * n,
       "u31": 0.5 * L + 16.0 * m_kept, "u21": 0.5 * L + 16.0 * m_kept}.get(leg)


def fn():
    rc = calls[leg]()
    assert rc == 0, ctx.last_error()


if args.once:
    if args.shifts:  # the first pair only: the two launches a PMC pass profiles, at a chosen place of the block
        sa, sb = (int(float(x)) for x in args.shifts.split(",")[0].split(":"))
        pa, pb = pa + sa, pb + sb
    fn()
    fn()
    torch.cuda.synchronize()
    ctx.sync()
    print(f"{args.leg} {args.alloc} once kept={m_kept}", flush=True)
    sys.exit(0)

if args.shifts:
    assert mode == "carve"
    pa0, pb0 = pa, pb
    for pair in args.shifts.split(","):
        sa, sb = (int(float(x)) for x in pair.split(":"))
        pa, pb = pa0 + sa, pb0 + sb
        with torch.cuda.stream(stream):
            for _ in range(6):
                fn()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
            ev[0].record(stream)
            for i in range(5):
                fn()
                ev[i + 1].record(stream)
            torch.cuda.synchronize()
        ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
        med = float(np.median(ts))
        print(f"{leg} shift a {sa:10d} b {sb:10d}: {med:.4f} ms frac {alg / med / 1e6 / 8000:.4f}", flush=True)
    sys.exit(0)

with torch.cuda.stream(stream):
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < args.busy_ms:
        fn()
        fn()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)]
    ev[0].record(stream)
    for i in range(args.reps):
        fn()
        ev[i + 1].record(stream)
    torch.cuda.synchronize()
rc, _ = ctx.sync()
assert rc == 0, ctx.last_error()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.reps)]
med = float(np.median(ts))
frac = f"frac {alg / med / 1e6 / 8000:.4f}" if alg else ""
nosplit = " split" if args.split else ""
shape = "x".join(str(v) for v in ctx.last_launch_shape())
print(f"{args.leg:7s} {args.alloc:12s} shape {shape} tile {args.tile:5d} thr {args.threads:3d} sub {args.subtiles}{nosplit}: {med:.4f} ms (min {min(ts):.4f} max {max(ts):.4f}) {frac} kept={m_kept} a at {pa:#x}", flush=True)
