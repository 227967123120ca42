#!/usr/bin/env python3
"""Where a workgroup of the canonical kernel spends its life: reads the s_memrealtime stamps of
a -DKMERS_STAMPS diagnostic build (tools/libkmers_stamps.so).  Shares, not absolute run time."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkmers_stamps.so"))
for name, (res, args) in cap.SYMBOLS.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
h = C.c_void_p(); assert lib.kmers_ctx_create(0, None, C.byref(h)) == 0
dev = torch.device("cuda", 0)
K, L = 31, 1_000_000_000
for bits, hashes, tile in ((4, True, 512), (4, True, 1024), (4, True, 2048), (2, True, 1024), (4, False, 2048), (8, True, 1024), (8, True, 512)):
    nw = (L * bits + 63) // 64; n = L - K + 1
    if bits == 8:
        buf = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=dev)[torch.randint(0, 4, (nw * 8 + 16,), device=dev)].view(torch.int64)
        torch.cuda.synchronize()
    else:
        buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # the fill runs on torch's stream, the generator on the library's
        assert lib.kmers_synth_dna(h, 1, 0, nw, bits, 0, buf.data_ptr()) == 0
    ok = torch.empty(n, dtype=torch.int64, device=dev)
    oh = torch.empty(n, dtype=torch.int64, device=dev) if hashes else None
    ntiles = (n + tile - 1) // tile
    st = torch.zeros(((ntiles >> 10) + 2) * 4 * 8, dtype=torch.int64, device=dev)
    lib.kmers_ctx_set_param(h, cap.PARAM_TILE_KMERS, tile)
    lib.kmers_ctx_set_param(h, 3, st.data_ptr())
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0); res = cap.Result()
    for _ in range(3):
        rc = lib.kmers_canonical(h, C.byref(seq), K, 2, ok.data_ptr(), oh.data_ptr() if hashes else None, 0, cap.MEM_DEVICE, C.byref(res))
        assert rc == 0, (rc, lib.kmers_last_error(h), res.err_pos, res.err_enc)
    s = st.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 0] != 0]
    d = np.diff(s[:, :5], axis=1) * 10.0  # 100 MHz ticks -> ns
    t_all = (s[:, 4].max() - s[:, 0].min()) * 10.0
    print(f"src_bits={bits} hashes={hashes} tile={tile}: waves sampled {len(s)}  kernel span {t_all / 1e6:.3f} ms")
    for name, col in (("entry->loads+convert done", 0), ("barrier", 1), ("phase2 (LDS, ALU, store issue)", 2), ("store drain (vmcnt 0)", 3)):
        print(f"    {name:32s} mean {d[:, col].mean():8.0f} ns   median {np.median(d[:, col]):8.0f}   p90 {np.percentile(d[:, col], 90):8.0f}")
    print(f"    {'total':32s} mean {d.sum(axis=1).mean():8.0f} ns")
    del buf, ok, oh, st
