#!/usr/bin/env python3
"""UnambiguousKmers rates (1 Gbase LongDNA{4}) over tile sizes and library variants: whole-call time of the device-pointer path."""
import argparse
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
ap = argparse.ArgumentParser()
ap.add_argument("--libs", default=cap.library_path())
ap.add_argument("--tiles", default="0")
ap.add_argument("--cases", default="k31,c5,clean")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--grids", default="0")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = 1_000_000_000
nw = (L * 4 + 63) // 64


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in cap.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


libs = [(os.path.basename(p), load(p)) for p in a.libs.split(",")]
ctxs = []
for name, lib in libs:
    h = C.c_void_p()
    assert lib.kmers_ctx_create(0, None, C.byref(h)) == 0
    ctxs.append(h)
lib0, h0 = libs[0][1], ctxs[0]
amb = torch.empty(nw + 2, dtype=torch.int64, device=dev)
clean = torch.empty(nw + 2, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
assert lib0.kmers_synth_dna(h0, 7, 0, nw, 4, 2621, amb.data_ptr()) == 0
assert lib0.kmers_synth_dna(h0, 7, 0, nw, 4, 0, clean.data_ptr()) == 0
kk = torch.empty(L, dtype=torch.int64, device=dev)
ss = torch.empty(L, dtype=torch.int64, device=dev)
res = cap.Result()
cases = {"k31": (amb, 31, 1), "c5": (amb, 21, 3), "clean": (clean, 31, 1)}
for case in a.cases.split(","):
    buf, K, J = cases[case]
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
    for (name, lib), h in zip(libs, ctxs):
        for t, g in [(t, g) for t in a.tiles.split(",") for g in a.grids.split(",")]:
            lib.kmers_ctx_set_param(h, cap.PARAM_TILE_KMERS, int(t))
            lib.kmers_ctx_set_param(h, cap.PARAM_MAX_GRID, int(g))
            stream = torch.cuda.ExternalStream(lib.kmers_ctx_stream(h), device=dev)
            ts = []
            for r in range(a.reps + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record(stream)
                rc = lib.kmers_unambiguous(h, C.byref(seq), K, J, kk.data_ptr(), ss.data_ptr(), L, cap.MEM_DEVICE, C.byref(res))
                e1.record(stream)
                torch.cuda.synchronize()
                assert rc == 0, rc
                if r:
                    ts.append(e0.elapsed_time(e1))
            m = int(res.n_out)
            med = float(np.median(ts))
            alg = 0.5 * L + 16.0 * m
            print(f"{case:6s} {name:24s} tile {int(t):5d} grid {int(g):5d}  {med:7.3f} ms (min {min(ts):.3f})  kept {m}  {alg / med / 1e6:7.1f} GB/s = {alg / med / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
