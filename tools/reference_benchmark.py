#!/usr/bin/env python3
"""The reference's own throughput smoke test (test/benchmark.jl:35-123) on the GPU, with the CPU
restatement beside it: 10 M symbols, K = 7, every iterator reduced by XOR over `kmer.data[1]`
(test/benchmark.jl:9-15), plus the minimizer loop (K = 8, W = 20, every 20th start, :96-119).

Sources as in the reference: a 2-bit DNA LongSequence, a 4-bit RNA LongSequence (its iterators build
4-bit kmers: `FwKmers{typeof(Alphabet(seq)), 7}`), and a String over "AaCcGgTt".  (The amino-acid
rows are out of scope)  Every iterator row uses the fused XOR consumer (`kmers_reduce_xor_iter`); the minimizer loop
materialises its elements in HBM and folds them there.  Every GPU value is checked against the
oracle's.  `--n` changes the length (the reference uses 10 M)."""
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
ap.add_argument("--n", type=int, default=10_000_000)
args = ap.parse_args()
N = args.n
cap = km._capi
ctx = km.Context(0)
orc = pyoracle.Oracle(pyoracle.build(native=True))
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
res = cap.Result()
K = 7


def fold(t):
    while t.numel() > 1:
        h = t.numel() // 2
        rest = t[2 * h:]
        t = torch.bitwise_xor(t[:h], t[h:2 * h])
        if rest.numel():
            t = torch.cat([t, rest])
    return int(t.item()) & (2**64 - 1) if t.numel() else 0


def best_of(fn, reps=5, torch_work=True):
    """torch_work: the case also runs torch kernels (the XOR fold), so bracket it with device syncs;
    the C ABI calls themselves are synchronous."""
    out, best = None, 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        if torch_work:
            torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return out, best


def cpu_time(fn):
    t0 = time.perf_counter()
    out = fn()
    return out, time.perf_counter() - t0


sources = {}
rng = np.random.default_rng(439824)
for name, bits in (("2-bit LongSequence", 2), ("4-bit LongSequence", 4)):
    nw = (N * bits + 63) // 64
    host = orc.synth_words(bits, 0, nw + 1, bits)
    d = torch.from_numpy(host.view(np.int64).copy()).to(dev)
    sources[name] = (host, d, bits, bits)  # oracle source code, kmer bits = source bits (typeof(Alphabet(seq)))
text = rng.choice(np.frombuffer(b"AaCcGgTt", np.uint8), N)
host = np.zeros((N + 7) // 8 + 1, np.uint64)
host.view(np.uint8)[:N] = text
sources["String"] = (host, torch.from_numpy(host.view(np.int64).copy()).to(dev), 8, 2)
torch.cuda.synchronize()

rows = []


def row(section, name, gpu, cpu, ok):
    rows.append((section, name, gpu, cpu, ok))
    print(f"{section:28s} {name:20s} GPU {gpu * 1e3:9.3f} ms   CPU (1 thread) {cpu * 1e3:10.2f} ms   x{cpu / gpu:8.0f}   {'ok' if ok else 'MISMATCH'}", flush=True)


def seq_of(d, src_bits):
    return cap.Seq(d.data_ptr(), N, 0, 0, src_bits, 0)


with torch.cuda.stream(stream):
    for section, canonical in (("FwKmers", 0), ("FwRvIterator (first)", 0), ("CanonicalKmers", 1)):
        for name, (host, d, src, dst) in sources.items():
            seq = seq_of(d, src)
            val = C.c_uint64()

            def gpu():
                ctx.check(ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, dst, canonical, C.byref(val), cap.MEM_DEVICE, C.byref(res)), "xor")
                return val.value
            g, tg = best_of(gpu, torch_work=False)
            if canonical:
                (c, _), tc = cpu_time(lambda: orc.reduce_xor_canonical(host, N, src, dst, K))
            else:
                (kmers, _), tc = cpu_time(lambda: orc.fw_kmers(host, N, src, dst, K))
                c = int(np.bitwise_xor.reduce(kmers[:, 0]))
            row(section, name, tg, tc, g == c)

    for name, (host, d, src, dst) in sources.items():
        seq = seq_of(d, src)
        val = C.c_uint64()

        def gpu():
            ctx.check(ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_UNAMBIGUOUS, 1, C.byref(val), cap.MEM_DEVICE,
                                                    C.byref(res)), "xor unambiguous")
            return val.value
        g, tg = best_of(gpu, torch_work=False)
        (kmers, _, _), tc = cpu_time(lambda: orc.unambiguous(host, N, src, K))
        row("UnambiguousKmers", name, tg, tc, g == int(np.bitwise_xor.reduce(kmers[:, 0])))

    for J in (5, 7):
        for name, (host, d, src, dst) in sources.items():
            seq = seq_of(d, src)
            val = C.c_uint64()

            def gpu():
                ctx.check(ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, dst, cap.ITER_SPACED, J, C.byref(val), cap.MEM_DEVICE,
                                                        C.byref(res)), "xor spaced")
                return val.value
            g, tg = best_of(gpu, torch_work=False)
            (kmers, _), tc = cpu_time(lambda: orc.spaced(host, N, src, dst, K, J))
            row(f"SpacedKmers, step {J}", name, tg, tc, g == int(np.bitwise_xor.reduce(kmers[:, 0])))

    host, d, src, dst = sources["2-bit LongSequence"]
    seq = seq_of(d, src)
    Km, W, step = 8, 20, 20
    n = (N - (Km + W - 1)) // step + 1
    out_k = torch.empty(n, dtype=torch.int64, device=dev)

    def gpu():
        ctx.check(ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), Km, W, step, 2, 0, out_k.data_ptr(), cap.MEM_DEVICE, C.byref(res)), "minimizers")
        return fold(out_k)
    g, tg = best_of(gpu)
    (mins, _), tc = cpu_time(lambda: orc.minimizers(host, N, src, 2, Km, W, step, 0))
    row("Minimizer (K=8, W=20)", "2-bit LongSequence", tg, tc, g == int(np.bitwise_xor.reduce(mins[:, 0])))

print(f"\n{N} symbols per case; GPU = best of 5 wall-clock calls through the C ABI (synchronous, data resident in HBM); "
      f"CPU = the C restatement of Kmers.jl (oracle/, gcc -O3 -march=native), one thread, including its output arrays.")
if not all(r[4] for r in rows):
    sys.exit("MISMATCH between GPU and oracle")
