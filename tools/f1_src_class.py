#!/usr/bin/env python3
"""f1 (the headline launch from 1 Gbase of ASCII text: 1 GB read, 16 GB written) with the TEXT in a block of the class pool too:
does it matter which region class the read stream lies in beside the two write streams?  Blocks are allocated in an order that
makes the pool put the text into the kmers' class, the hashes' class, or the third.    python3 tools/f1_src_class.py"""
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
L, K = 1_000_000_000, 31
n = L - K + 1
res = cap.Result()
ASYNC = cap.MEM_DEVICE | cap.ASYNC


class Raw:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


def run(order):
    ptr = {}
    for name in order:
        ptr[name] = ctx.alloc({"a": 8 * n, "b": 8 * n, "t": max(L + 64, 1 << 30)}[name])
    lay = {k: "".join("ABC?"[c] for c in ctx.pool_layout(v)[1]) for k, v in ptr.items()}
    with torch.cuda.stream(stream):
        text = torch.as_tensor(Raw(ptr["t"], L + 64), device=dev)
        idx = torch.randint(0, 4, (L,), dtype=torch.uint8, device=dev)
        text.fill_(65)
        for code, add in ((1, 2), (2, 6), (3, 19)):
            text[:L] += idx.eq(code).to(torch.uint8) * add
        del idx
        seq = cap.Seq(ptr["t"], L, 0, 0, 8, 0)
        call = lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, ptr["a"], ptr["b"], 0, ASYNC, C.byref(res))
        for _ in range(40):
            call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        ev[0].record(stream)
        for i in range(10):
            call()
            ev[i + 1].record(stream)
        torch.cuda.synchronize()
    ms = float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(10)]))
    print(f"order {''.join(order)}: kmers {lay['a']} hashes {lay['b']} text {lay['t']}: {ms:.4f} ms = {17.0 * n / ms / 1e6 / 8000:.4f} of 8 TB/s", flush=True)
    del text
    for v in ptr.values():
        ctx.free(v)


for order in (("a", "b", "t"), ("t", "a", "b"), ("a", "t", "b"), ("b", "t", "a")):
    run(order)
