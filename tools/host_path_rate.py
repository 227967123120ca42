#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry (KMERS_MEM_HOST): H2D of the words, kernel,
D2H of kmers + hashes.  Reported in DESIGN.md next to the resident-in-HBM figure; never the
headline value."""
import ctypes as C
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import kmers_jl_amd as km
from oracle import pyoracle

cap = km._capi
ctx = km.Context(0)
orc = pyoracle.get()
K, bits = 31, 4
for L in (16_000_000, 256_000_000):
    nw = (L * bits + 63) // 64
    words = orc.synth_words(11, 0, nw + 1, bits)
    n = L - K + 1
    kmers = np.zeros(n, dtype=np.uint64)
    hashes = np.zeros(n, dtype=np.uint64)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, bits, 0)
    res = cap.Result()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, kmers.ctypes.data_as(C.c_void_p),
                                     hashes.ctypes.data_as(C.c_void_p), 0, cap.MEM_HOST, C.byref(res))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    print(f"host-pointer kmers_canonical L={L}: {best * 1e3:.1f} ms  {L / best / 1e9:.3f} Gbases/s "
          f"({(nw * 8 + n * 16) / best / 1e9:.1f} GB/s over PCIe, pageable host memory)")
