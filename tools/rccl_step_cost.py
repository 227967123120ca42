#!/usr/bin/env python3
"""What one grouped ncclSend + ncclRecv of the halo's size costs on the context's stream: kmers_comm_sendrecv with the rank as
its own peer (all a 1-GPU box admits), alone and in front of the C2 launch."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import kmers_jl_amd as km
from kmers_jl_amd.shard import NativeComm
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
comm = NativeComm.create(ctx, NativeComm.new_id(ctx.lib), 1, 0)
a = torch.zeros(16, dtype=torch.int64, device=dev)
b = torch.zeros(16, dtype=torch.int64, device=dev)
L, K = 1_000_000_000, 31
nw = L // 16 + 2
src = torch.empty(nw, dtype=torch.int64, device=dev)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 4, 0, nw - 2, 4, 0, src.data_ptr()), "synth")
seq = cap.Seq(src.data_ptr(), L, 0, 0, 4, 0)
ok = torch.empty(L, dtype=torch.int64, device=dev)
oh = torch.empty(L, dtype=torch.int64, device=dev)
res = cap.Result()
F = cap.MEM_DEVICE | cap.ASYNC
torch.cuda.synchronize()


def timed(fn, n=50):
    for _ in range(5):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(stream)
    for i in range(n):
        fn(); ev[i + 1].record(stream)
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))


xchg = lambda: comm.sendrecv(a.data_ptr(), 2, 0, b.data_ptr(), 2, 0)
kern = lambda: ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, ok.data_ptr(), oh.data_ptr(), 0, F, C.byref(res)), "c")
print(f"grouped ncclSend + ncclRecv of 16 B (self): {timed(xchg) * 1e3:.1f} us per call on the stream")
t_k = timed(kern, 20)
t_both = timed(lambda: (xchg(), kern()), 20)
print(f"C2 launch alone {t_k:.4f} ms; exchange + launch {t_both:.4f} ms  (+{(t_both - t_k) * 1e3:.1f} us, {100 * (t_both / t_k - 1):.2f} %)")
comm.close()
