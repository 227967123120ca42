#!/usr/bin/env python3
"""Where a wavefront of run_kernel (the fused XOR reducer / MinHash candidates) spends its life: a -DKMERS_STAMPS build of
consumers_api.hip (python -m kmers_jl_amd.build variant rstamps -DKMERS_STAMPS consumers_api.hip), per wavefront: first and last
s_memrealtime, shader cycles waiting at the two barriers of a tile, staging, rolling."""
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
L = 1_000_000_000
nw = (L * 4 + 63) // 64
buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, 4, 0, buf.data_ptr()), "synth")
seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
res = cap.Result()
val = C.c_uint64()
for grid in [int(g) for g in (sys.argv[1] if len(sys.argv) > 1 else "2048,1024").split(",")]:
    ctx.set_param(cap.PARAM_MAX_GRID, grid)
    st = torch.zeros(grid * 4 * 8, dtype=torch.int64, device=dev)
    ctx.set_param(3, st.data_ptr())
    for _ in range(3):
        st.zero_()
        torch.cuda.synchronize()
        assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)) == 0
    s = st.cpu().numpy().reshape(-1, 8).astype(np.float64)
    s = s[s[:, 0] != 0]
    span = (s[:, 1].max() - s[:, 0].min()) * 10.0
    life = (s[:, 1] - s[:, 0]) * 10.0
    start = (s[:, 0] - s[:, 0].min()) * 10.0
    print(f"grid {grid}: {len(s)} wavefronts, kernel span {span / 1e6:.3f} ms; a wavefront lives {life.mean() / 1e6:.3f} ms (min {life.min() / 1e6:.3f}), "
          f"starts {start.mean() / 1e3:.1f} us after the first (max {start.max() / 1e3:.1f} us), {s[:, 6].mean():.1f} tiles")
    if os.environ.get("RUN_STAMPS_DETAIL"):
        full = st.cpu().numpy().reshape(-1, 4, 8).astype(np.float64)   # [workgroup][wave][slot]
        wl = (full[:, 0, 1] - full[:, 0, 0]) * 10.0 / 1e3                # lifetime of every workgroup's wave 0, us
        print("    lifetime percentiles (us):", " ".join(f"{np.percentile(wl, q):.0f}" for q in (0, 10, 25, 50, 75, 90, 100)))
        b = np.arange(len(wl))
        print("    mean lifetime by blockIdx % 8 (XCD under round-robin):", " ".join(f"{wl[b % 8 == x].mean():.0f}" for x in range(8)))
        print("    mean lifetime by (blockIdx // 8) % 32:", " ".join(f"{wl[(b // 8) % 32 == x].mean():.0f}" for x in range(32)))
        print("    mean lifetime by blockIdx // 256:", " ".join(f"{wl[b // 256 == x].mean():.0f}" for x in range(len(wl) // 256)))
        for i, nm in enumerate(["barrier front", "stage", "barrier behind", "cut + roll"]):
            per = full[:, :, 2 + i].mean(axis=1) / full[:, :, 6].mean(axis=1)
            print(f"    cycles per tile by blockIdx // 256, {nm:15s}:", " ".join(f"{per[b // 256 == x].mean():.0f}" for x in range(len(wl) // 256)))
    tot = s[:, 2:6].sum(axis=1).mean()
    for i, nm in enumerate(["barrier in front of the stage", "stage (incl. waiting for the words)", "barrier behind the stage", "window cut + 32 rolling steps"]):
        print(f"    {nm:40s} {s[:, 2 + i].mean() / s[:, 6].mean():9.0f} cycles per tile  ({100 * s[:, 2 + i].mean() / tot:.0f} %)")
