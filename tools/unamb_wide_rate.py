import ctypes as C, os, sys, torch
sys.path.insert(0, "/root/repo")
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
dev = torch.device("cuda", 0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
L = 100_000_000
res = cap.Result()
with torch.cuda.stream(stream):
    nw = L * 4 // 64 + 1
    buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, 4, 131, buf.data_ptr()), "synth")   # p(N) = 0.002
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
    for K in (64, 128, 129, 150, 256, 1000):
        N = (2 * K + 63) // 64
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        a = torch.empty(max(m, 1) * N, dtype=torch.int64, device=dev)
        s = torch.empty(max(m, 1), dtype=torch.int64, device=dev)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, a.data_ptr(), s.data_ptr(), m, cap.MEM_DEVICE | cap.ASYNC, C.byref(res))
            e1.record(stream)
            torch.cuda.synchronize()
            ctx.sync()
            best = min(best, e0.elapsed_time(e1))
        gb = (m * (8 * N + 8) + L / 2) / 1e9
        print(f"K {K:5d} N {N:3d}: kept {m} {best:8.3f} ms {gb / best * 1e3:8.1f} GB/s", flush=True)
