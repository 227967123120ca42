#!/usr/bin/env python3
"""UnambiguousDNAMers{31} from ASCII text (1 Gbase, N at p = 0.04 / clean) against the same from LongDNA{4}: whole call, device outputs."""
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
res = cap.Result()
g = torch.Generator(device=dev); g.manual_seed(1)
letters = torch.tensor([65, 67, 71, 84, 97, 99, 103, 116], dtype=torch.uint8, device=dev)
text = letters[torch.randint(0, 8, (L + 64,), device=dev, generator=g)]
clean = text.clone()
text[torch.rand(L + 64, device=dev, generator=g) < 0.04] = 78  # 'N'
nw4 = L // 16 + 2
w4 = torch.empty(nw4, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw4 - 2, 4, 2621, w4.data_ptr()), "synth")
kk = torch.empty(L, dtype=torch.int64, device=dev)
ss = torch.empty(L, dtype=torch.int64, device=dev)
for label, seq, bpb in (("ASCII p(N)=0.04", cap.Seq(text.data_ptr(), L, 0, 0, 8, 0), 1.0), ("ASCII clean", cap.Seq(clean.data_ptr(), L, 0, 0, 8, 0), 1.0),
                        ("LongDNA{4} p(N)=0.04", cap.Seq(w4.data_ptr(), L, 0, 0, 4, 0), 0.5)):
    fn = lambda: ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, kk.data_ptr(), ss.data_ptr(), L, cap.MEM_DEVICE, C.byref(res)), label)
    fn(); fn()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    med = float(np.median(ts)); m = int(res.n_out)
    print(f"{label:24s} {med:.3f} ms  kept {m}  {(bpb * L + 16.0 * m) / med / 1e6 / 8000:.3f} of 8 TB/s on {bpb} B/base + 16 B/kept", flush=True)
