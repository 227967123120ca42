#!/usr/bin/env python3
"""CanonicalDNAMers{31} + fx_hash from ASCII text against the same from LongDNA{4}, interleaved in one process (kernel time)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
libs = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] else [cap.library_path()]
tiles = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
dev = torch.device("cuda", 0)
L, K = 1_000_000_000, 31


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in cap.SYMBOLS.items():
        fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
    return lib


text = torch.tensor([65, 67, 71, 84, 97, 99, 103, 116], dtype=torch.uint8, device=dev)[torch.randint(0, 8, (L + 64,), device=dev)]
ok, oh = torch.empty(L, dtype=torch.int64, device=dev), torch.empty(L, dtype=torch.int64, device=dev)
res = cap.Result()
variants = []
for p in libs:
    lib = load(p); h = C.c_void_p(); assert lib.kmers_ctx_create(0, None, C.byref(h)) == 0
    variants.append((os.path.basename(p), lib, h))
lib0, h0 = variants[0][1], variants[0][2]
w4 = torch.empty(L // 16 + 2, dtype=torch.int64, device=dev); torch.cuda.synchronize()
assert lib0.kmers_synth_dna(h0, 5, 0, L // 16, 4, 0, w4.data_ptr()) == 0
seqs = {"ASCII": cap.Seq(text.data_ptr(), L, 0, 0, 8, 0), "4-bit": cap.Seq(w4.data_ptr(), L, 0, 0, 4, 0)}
times = {}
F = cap.MEM_DEVICE | cap.ASYNC
for rnd in range(12):
  for tile in tiles:
    for name, lib, h in variants:
        lib.kmers_ctx_set_param(h, cap.PARAM_TILE_KMERS, tile)
        name = f"{name} tile {tile}"
        stream = torch.cuda.ExternalStream(lib.kmers_ctx_stream(h), device=dev)
        for label, seq in seqs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            assert lib.kmers_canonical(h, C.byref(seq), K, 2, ok.data_ptr(), oh.data_ptr(), 0, F, C.byref(res)) == 0
            e1.record(stream); torch.cuda.synchronize()
            if rnd >= 2:
                times.setdefault((name, label), []).append(e0.elapsed_time(e1))
for (name, label), ts in times.items():
    med = float(np.median(ts)); b = (1.0 if label == "ASCII" else 0.5) + 16.0
    print(f"{name:24s} {label:6s} {med:.4f} ms (min {min(ts):.4f})  {b * (L - K + 1) / med / 1e6 / 8000:.3f} of 8 TB/s")
