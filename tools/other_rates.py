#!/usr/bin/env python3
"""Kernel rates of the non-headline entry points on 1 Gbase (HIP events around each call)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
L = 1_000_000_000


def empty(n):
    return torch.empty(int(n), dtype=torch.int64, device=dev)


def timed(label, fn, bytes_total, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = float(np.median(ts))
    print(f"{label:62s} {t:8.3f} ms  {L / t / 1e6:7.1f} Gbases/s  {bytes_total / t / 1e6:7.1f} GB/s ({bytes_total / t / 1e6 / 80:.1f}% of 8 TB/s)")


res = cap.Result()
for bits in (4, 2):
    nw = (L * bits + 63) // 64
    buf = empty(nw + 2); torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, bits, 0, buf.data_ptr()), "synth")
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    r = bits / 8
    # spaced (C5 strict shape)
    K, J = 21, 3
    n = (L - K) // J + 1
    out = empty(n)
    timed(f"src={bits} SpacedDNAMers{{21,3}} strict", lambda: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, out.data_ptr(), cap.MEM_DEVICE, C.byref(res)), L * r + n * 8)
    # unambiguous on a clean sequence (every window kept): count + scan + emit
    K = 31
    n = L - K + 1
    kk, ss = empty(n), empty(n)
    timed(f"src={bits} UnambiguousDNAMers{{31}} clean (kmer+start)", lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, kk.data_ptr(), ss.data_ptr(), n, cap.MEM_DEVICE, C.byref(res)), 2 * L * r + n * 16)
    del out, kk, ss
    if bits == 4:
        amb = empty(nw + 2); torch.cuda.synchronize()
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, bits, 2621, amb.data_ptr()), "synth")
        seqa = cap.Seq(amb.data_ptr(), L, 0, 0, bits, 0)
        ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), 31, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res))
        m = int(res.n_out)
        kk, ss = empty(m), empty(m)
        timed(f"src=4 UnambiguousDNAMers{{31}} p(N)=0.04 ({m / n:.3f} kept)", lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), 31, 1, kk.data_ptr(), ss.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)), 2 * L * r + m * 16)
        ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), 21, 3, None, None, 0, cap.MEM_DEVICE, C.byref(res))
        m = int(res.n_out)
        timed(f"src=4 Spaced{{21,3}} skip variant (C5) ({m} kept)", lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), 21, 3, kk.data_ptr(), ss.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)), 2 * L * r + m * 16)
        del kk, ss, amb
    # batch ops on 1e9 kmers
    n = L - 31 + 1
    a, b = empty(n), empty(n)
    ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 31, 2, a.data_ptr(), None, cap.MEM_DEVICE, C.byref(res))
    if bits == 4:
        timed("fx_hash over 1e9 one-word kmers", lambda: ctx.lib.kmers_fx_hash(ctx.handle, a.data_ptr(), 1, n, 0, b.data_ptr(), cap.MEM_DEVICE), n * 16)
        timed("reverse_complement over 1e9 one-word kmers", lambda: ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, a.data_ptr(), 31, 2, n, b.data_ptr(), cap.MEM_DEVICE), n * 16)
        timed("canonical over 1e9 one-word kmers", lambda: ctx.lib.kmers_transform(ctx.handle, cap.OP_CANONICAL, a.data_ptr(), 31, 2, n, b.data_ptr(), cap.MEM_DEVICE), n * 16)
    del a, b, buf
# ASCII source
raw = empty(L // 8 + 2)
torch.cuda.synchronize()
host = np.frombuffer(np.random.default_rng(1).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=1 << 24).tobytes(), dtype=np.uint8)
t = torch.from_numpy(np.tile(host, L // (1 << 24) + 1)[:L + 16].copy()).to(dev)
seq = cap.Seq(t.data_ptr(), L, 0, 0, 8, 0)
n = L - 30
ck, hs = empty(n), empty(n)
timed("src=ASCII CanonicalDNAMers{31} + fx_hash", lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), 31, 2, ck.data_ptr(), hs.data_ptr(), 0, cap.MEM_DEVICE, C.byref(res)), L * 1 + n * 16)
