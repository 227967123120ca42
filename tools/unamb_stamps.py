#!/usr/bin/env python3
"""Where a workgroup of the single-pass UnambiguousKmers kernel spends its life: s_memrealtime stamps of a -DKMERS_STAMPS
diagnostic build (tools/libkmers_stamps.so), every 16th tile, per wavefront."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
cap = km._capi
lib = C.CDLL(os.environ.get("KMERS_STAMPS_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libkmers_stamps.so"))
for name, (res, args) in cap.SYMBOLS.items():
    fn = getattr(lib, name); fn.restype, fn.argtypes = res, args
h = C.c_void_p(); assert lib.kmers_ctx_create(0, None, C.byref(h)) == 0
dev = torch.device("cuda", 0)
L = 1_000_000_000
nw = (L * 4 + 63) // 64
kk = torch.empty(L, dtype=torch.int64, device=dev); ss = torch.empty(L, dtype=torch.int64, device=dev)
names = ["ticket", "own words staged", "barrier (all staged)", "resolve", "look-back (wave 0)", "barrier (base known)", "emit (stores issued)", "store drain"]
CASES = (("k31 p(N)=0.04", 2621, 31, 1), ("c5 skip", 2621, 21, 3), ("clean", 0, 31, 1))
for label, amb, K, J in CASES[:int(os.environ.get("KMERS_STAMPS_CASES", "3"))]:
    buf = torch.empty(nw + 2, dtype=torch.int64, device=dev); torch.cuda.synchronize()
    assert lib.kmers_synth_dna(h, 7, 0, nw, 4, amb, buf.data_ptr()) == 0
    for tile in [int(t) for t in os.environ.get("KMERS_STAMPS_TILES", "32768,8192").split(",")]:
        ntiles = (L - K + 1 + tile - 1) // tile
        st = torch.zeros(((ntiles >> 4) + 2) * 4 * 16, dtype=torch.int64, device=dev)
        lib.kmers_ctx_set_param(h, cap.PARAM_TILE_KMERS, tile)
        lib.kmers_ctx_set_param(h, 3, st.data_ptr())
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0); res = cap.Result()
        for _ in range(3):
            st.zero_(); torch.cuda.synchronize()
            rc = lib.kmers_unambiguous(h, C.byref(seq), K, J, kk.data_ptr(), ss.data_ptr(), L, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0, rc
        s = st.cpu().numpy().reshape(-1, 16)
        s = s[s[:, 0] != 0]
        d = np.diff(s[:, :9], axis=1) * 10.0  # 100 MHz ticks -> ns
        span = (s[:, 8].max() - s[:, 0].min()) * 10.0
        print(f"{label}, tile {tile}: wavefronts sampled {len(s)}, kernel span {span / 1e6:.3f} ms")
        for i, nm in enumerate(names):
            print(f"    {nm:26s} mean {d[:, i].mean():8.0f} ns   median {np.median(d[:, i]):8.0f}   p90 {np.percentile(d[:, i], 90):8.0f}")
        print(f"    {'total':26s} mean {d.sum(axis=1).mean():8.0f} ns")
        for i, nm in enumerate(["back: listing", "back: whole frames", "back: carry", "back: round scan", "back: partial frames"]):  # shader cycles (s_memtime)
            print(f"    {nm:26s} mean {s[:, 10 + i].mean():8.0f} cycles")
        # start time of a tile against its ticket: how many are in flight
        t0 = s[:, 0].astype(np.float64) * 10.0; t0 -= t0.min()
        order = np.argsort(s[:, 9])
        print(f"    tiles start in ticket order: corr(tile id, start time) = {np.corrcoef(s[order, 9], t0[order])[0, 1]:.4f}")
    del buf
