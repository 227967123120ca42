#!/usr/bin/env python3
"""Runs the BASELINE.json configs C1..C5 (plus the fused consumers) a few times each with resident
data, for `rocprofv3 --kernel-trace --stats -- python3 tools/run_configs.py`.  Prints the
algorithmic bytes per launch so the profile summary can be turned into GB/s."""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
res = cap.Result()
GOLDEN = 0x9E3779B97F4A7C15
REPS = 5


def empty(n):
    return torch.empty(int(n), dtype=torch.int64, device=dev)


def synth(seed, nw, bits, amb=0):
    b = empty(nw + 2)
    torch.cuda.synchronize()
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, b.data_ptr()), "synth")
    return b


info = {}
# C1
L, K, bits = 1_000_000, 21, 4
buf = synth(GOLDEN ^ 1, L * bits // 64 + 1, bits); out = empty(L)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
for _ in range(REPS):
    ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, out.data_ptr(), None, cap.MEM_DEVICE, C.byref(res))
info["C1 FwDNAMers{21} 1 Mbase LongDNA{4}"] = {"kernel": "stream_kernel<4, 2, 1, 0, true, false>", "bytes": 8.5 * (L - K + 1)}
# C2
L, K, bits = 1_000_000_000, 31, 4
buf = synth(GOLDEN ^ 2, L * bits // 64 + 1, bits); a = empty(L); b = empty(L)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
for _ in range(REPS):
    ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), 0, cap.MEM_DEVICE, C.byref(res))
info["C2 CanonicalDNAMers{31}+fx_hash 1 Gbase LongDNA{4}"] = {"kernel": "stream_kernel<4, 2, 1, 1, true, false>", "bytes": 16.5 * (L - K + 1)}
# C3 per-GPU share: 1.25 Gbase of LongDNA{2}, canonical only
L, K, bits = 1_250_000_000, 31, 2
buf = synth(GOLDEN ^ 3, L * bits // 64 + 1, bits); a = empty(L)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
for _ in range(REPS):
    ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, cap.MEM_DEVICE, C.byref(res))
info["C3 CanonicalDNAMers{31} 1.25 Gbase LongDNA{2} (one of 8 shards)"] = {"kernel": "stream_kernel<2, 2, 1, 1, true, false>", "bytes": 8.25 * (L - K + 1)}
# C4
L, K, bits = 1_000_000_000, 63, 4
buf = synth(GOLDEN ^ 4, L * bits // 64 + 1, bits); a = empty(2 * L); b = empty(2 * L)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
for _ in range(REPS):
    ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), cap.MEM_DEVICE, C.byref(res))
info["C4 FwDNAMers{63}+reverse_complement 1 Gbase LongDNA{4}"] = {"kernel": "stream_kernel<4, 2, 2, 0, true, false>", "bytes": 32.5 * (L - K + 1)}
del b
# C5 strict + skip
L, K, J, bits = 1_000_000_000, 21, 3, 4
n = (L - K) // J + 1
buf = synth(GOLDEN ^ 5, L * bits // 64 + 1, bits)
seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
for _ in range(REPS):
    ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, a.data_ptr(), cap.MEM_DEVICE, C.byref(res))
info["C5 SpacedDNAMers{21,3} strict 1 Gbase LongDNA{4}"] = {"kernel": "stream_kernel<4, 2, 1, 0, false, false>", "bytes": 0.5 * L + 8.0 * n}
amb = synth(GOLDEN ^ 5, L * bits // 64 + 1, bits, 2621)
seqa = cap.Seq(amb.data_ptr(), L, 0, 0, bits, 0)
ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res))
m = int(res.n_out)
s = empty(m)
for _ in range(REPS):
    ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, a.data_ptr(), s.data_ptr(), m, cap.MEM_DEVICE, C.byref(res))
info["C5 skip variant (UnambiguousDNAMers{21} on the stride-3 lattice), p(N)=0.04"] = {
    "kernel": "unambiguous_kernel<4, 1, false> + scan + unambiguous_kernel<4, 1, true>", "bytes": 2 * 0.5 * L + 16.0 * m, "kept": m}
# fused consumers on the C2 input
xr = C.c_uint64()
for _ in range(REPS):
    ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res))
info["fused XOR-reduce canonical K=31"] = {"kernel": "stream_kernel<4, 2, 1, 2, true, false>", "bytes": 0.5 * L}
print(json.dumps(info, indent=1))
