#!/usr/bin/env python3
"""The element-wise legs (f4: kmers_fx_hash / reverse_complement over an array of 1 G one-word kmers, 8 B read + 8 B written per
element) with source and destination at chosen places of ONE 200 GiB block: the copy ceiling of the device as a function of
placement (profiles/r04_copy.md).    python3 tools/copy_leg.py --op hash|revcomp --dst-gib 8,32,64,128 [--once]"""
import argparse
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km

ap = argparse.ArgumentParser()
ap.add_argument("--op", default="hash")
ap.add_argument("--dst-gib", default="8,16,32,48,64,96,128,160,188")
ap.add_argument("--once", action="store_true", help="two launches per placement and nothing else (PMC passes)")
args = ap.parse_args()
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
n = 1_000_000_000
block = torch.empty(200 << 27, dtype=torch.int64, device=dev)
base = block.data_ptr()
block[: n].random_(0, 1 << 62)
torch.cuda.synchronize()
ASYNC = cap.MEM_DEVICE | cap.ASYNC
for g in [int(x) for x in args.dst_gib.split(",")]:
    dst = base + (g << 30)
    if args.op == "hash":
        fn = lambda: ctx.lib.kmers_fx_hash(ctx.handle, C.c_void_p(base), 1, n, 0, C.c_void_p(dst), ASYNC)
    else:
        fn = lambda: ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, C.c_void_p(base), 31, 2, n, C.c_void_p(dst), ASYNC)
    with torch.cuda.stream(stream):
        if args.once:
            assert fn() == 0 and fn() == 0
            torch.cuda.synchronize()
            print(f"{args.op} destination at +{g} GiB: two launches", flush=True)
            continue
        for _ in range(20):
            assert fn() == 0
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
        ev[0].record(stream)
        for i in range(9):
            assert fn() == 0
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(9)]
    med = float(np.median(ts))
    print(f"{args.op} destination at +{g:3d} GiB: {med:.4f} ms  {16.0 * n / med / 1e9:.3f} TB/s  frac {16.0 * n / med / 1e6 / 8000:.4f}", flush=True)
