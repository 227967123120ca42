#!/usr/bin/env python3
"""The fused consumers of bench.py's other_configs, three launches each over 1 Gbase LongDNA{4}: the program rocprofv3 --pmc
profiles for tools/r2_fused_pmc.sh (instruction counts against the integer issue rate)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L = 1_000_000_000
nw = L // 16
d = ctx.alloc(nw * 8 + 16)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 42, 0, nw, 4, 0, d), "synth")
seq = cap.Seq(d, L, 0, 0, 4, 0)
res = cap.Result()
val = C.c_uint64()
out = np.zeros(1000, dtype=np.uint64)
c4, c8 = ctx.alloc(4 ** 4 * 4), ctx.alloc(4 ** 8 * 4)
for _ in range(3):
    ctx.check(ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)), "xor")
    ctx.check(ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, out.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)), "sketch")
    ctx.check(ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 4, c4, cap.MEM_DEVICE, C.byref(res)), "comp4")
    ctx.check(ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 8, c8, cap.MEM_DEVICE, C.byref(res)), "comp8")
