#!/usr/bin/env python3
"""Rate of the run-time-width kernels (kmers of more than four words) beside the widest tile-kernel case: CanonicalKmers + fx_hash,
FwKmers + reverse complements, the fused XOR reducer, over `--bases` of a 2-bit / 4-bit / text source.  Device-resident, HIP events."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kmers_jl_amd as km  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bases", type=int, default=100_000_000)
ap.add_argument("--ks", default="128,129,150,256,1000")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--arena", "--pool", dest="arena", action="store_true", help="outputs from kmers_dev_alloc (the class pool); the single-output launch also into a block taken by role")
ap.add_argument("--force-tiles", type=int, default=0, help="KMERS_PARAM_WIDE_NO_TILES value (2: the tile form for kmers of one to four words too)")
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
ctx.set_param(11, args.force_tiles)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
L = args.bases
res = cap.Result()
ASYNC = cap.MEM_DEVICE | cap.ASYNC


def timed(fn):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        rc = fn()
        e1.record(stream)
        torch.cuda.synchronize()
        assert rc == 0, ctx.last_error()
        best = min(best, e0.elapsed_time(e1))
    rc, _ = ctx.sync()
    assert rc == 0, ctx.last_error()
    return best


with torch.cuda.stream(stream):
    for src in (2, 4, 8):
        if src == 8:
            idx = torch.randint(0, 4, (L,), dtype=torch.uint8, device=dev)
            buf = torch.full((L + 16,), 65, dtype=torch.uint8, device=dev)
            for code, add in ((1, 2), (2, 6), (3, 19)):
                buf[:L] += idx.eq(code).to(torch.uint8) * add
            del idx
        else:
            nw = (L * src + 63) // 64
            buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
            ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, src, 0, buf.data_ptr()), "synth")
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, src, 0)
        for K in [int(x) for x in args.ks.split(",")]:
            N = (2 * K + 63) // 64
            n = L - K + 1
            if args.arena:
                pa, pb = ctx.alloc(n * N * 8), ctx.alloc(n * N * 8)

                class _P:
                    def __init__(self, p): self.p = p
                    def data_ptr(self): return self.p
                a, b = _P(pa), _P(pb)
            else:
                a = torch.empty(n * N, dtype=torch.int64, device=dev)
                b = torch.empty(n * N, dtype=torch.int64, device=dev)
            val = C.c_uint64()
            t_c = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, ASYNC, C.byref(res)))
            t_f = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), ASYNC, C.byref(res)))
            t_x = timed(lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
            t_1 = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, ASYNC, C.byref(res)))
            t_r = float("nan")
            if args.arena:
                ctx.free(pa)
                ctx.free(pb)
                pl = ctx.alloc(n * N * 8, lone_output=True)
                t_r = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, pl, None, ASYNC, C.byref(res)))
                t_cr = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, pl, None, 0, ASYNC, C.byref(res)))
                ctx.free(pl)
                print(f"   fw only {t_1:8.3f} ms ({8 * N * n / t_1 / 1e6:7.1f} GB/s), by role {t_r:8.3f} ms ({8 * N * n / t_r / 1e6:7.1f} GB/s); canonical kmers only by role {t_cr:8.3f} ms ({8 * N * n / t_cr / 1e6:7.1f} GB/s)")
            gb_c, gb_f = (8 * N + 8) * n / 1e9, 16 * N * n / 1e9
            print(f"src {src} K {K:5d} N {N:3d}: canonical+hash {t_c:9.3f} ms ({gb_c / t_c * 1e3:7.1f} GB/s)  fw+rc {t_f:9.3f} ms ({gb_f / t_f * 1e3:7.1f} GB/s)"
                  f"  xor-reduce {t_x:9.3f} ms ({n / t_x / 1e6:8.2f} G kmers/s)", flush=True)
            if not args.arena:
                del a, b
        del buf
