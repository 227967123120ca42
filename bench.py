#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer iteration hot path on MI355X.

Workload (BASELINE.json configs[1]): CanonicalDNAMers{31} + fx_hash over 1 Gbase of synthetic
uniform LongDNA{4} per GPU, materialising both the canonical kmers and their hashes
(16.5 algorithmic bytes per kmer: 0.5 read + 8 + 8 written).  A "step" is one pass of the hot
path over the rank's shard: the (K-1)-base halo exchange with the next rank (N > 1 only, RCCL)
followed by the canonical+hash kernel, inputs and outputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Weak scaling: every rank owns `--bases` symbols of one long
sequence of N * bases symbols; value = all symbols processed / max-over-ranks time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

METRIC = "canonical k-mers/sec (Gbases/s input) + % HBM roofline, K=31 DNA{4}"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy
FX_CONSTANT = 0x517CC1B727220A95


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def xor_fold(t):
    """XOR of all elements of an int64 CUDA tensor."""
    while t.numel() > 1:
        h = t.numel() // 2
        rest = t[2 * h:]
        t = torch.bitwise_xor(t[:h], t[h:2 * h])
        if rest.numel():
            t = torch.cat([t, rest])
    return int(t.item()) & (2**64 - 1)


def cpu_baseline(k, bits, seed, total_bases, budget_s=12.0, chunk_bases=1 << 24):
    """The oracle (C restatement of CanonicalKmers.jl:131-144 + kmer.jl:255-261, -O3 -march=native)
    timed on the host cores over a bounded sample of the same workload."""
    from oracle import pyoracle
    orc = pyoracle.Oracle(pyoracle.build(native=True))
    per_word = 64 // bits
    n_chunks = max(1, total_bases // chunk_bases)
    km = np.zeros((chunk_bases, 1), dtype=np.uint64)
    hs = np.zeros(chunk_bases, dtype=np.uint64)
    done, spent = 0, 0.0
    for c in range(n_chunks):
        words = orc.synth_words(seed, c * chunk_bases // per_word, chunk_bases // per_word + 1, bits)
        t0 = time.perf_counter()
        _, _, res = orc.canonical(words, chunk_bases, bits, 2, k, seed=0, out=km, out_h=hs)
        spent += time.perf_counter() - t0
        assert res.status == 0
        done += chunk_bases
        if spent >= budget_s:
            break
    one = {"value": round(done / spent / 1e9, 4), "unit": "Gbases/s", "cores": 1, "kind": "port",
           "sample": f"{done / 1e6:.0f} Mbase of the same synthetic LongDNA{{{bits}}} in {chunk_bases >> 20} Mi-base chunks, "
                     f"CanonicalDNAMers{{{k}}} + fx_hash materialised, 1 thread (the reference is single-threaded), "
                     f"C restatement of Kmers.jl (Julia absent from the image), gcc -O3 -march=native"}
    # all host cores: contiguous 1 Mi-base chunks, one stream per thread (ctypes releases the GIL)
    try:
        from concurrent.futures import ThreadPoolExecutor
        ncpu = len(os.sched_getaffinity(0))
        small = 1 << 20
        bufs = [(np.zeros((small, 1), np.uint64), np.zeros(small, np.uint64)) for _ in range(ncpu)]
        inputs = [orc.synth_words(seed, c * small // per_word, small // per_word + 1, bits) for c in range(ncpu)]
        rate1 = done / spent
        reps = max(1, int(budget_s / 4.0 * rate1 / small))  # about a quarter of the budget per thread (3 s by default)

        def work(i):
            for _ in range(reps):
                orc.canonical(inputs[i], small, bits, 2, k, seed=0, out=bufs[i][0], out_h=bufs[i][1])
        t0 = time.perf_counter()
        with ThreadPoolExecutor(ncpu) as ex:
            list(ex.map(work, range(ncpu)))
        dt = time.perf_counter() - t0
        one["all_cores"] = {"value": round(ncpu * reps * small / dt / 1e9, 4), "unit": "Gbases/s", "cores": ncpu,
                            "sample": f"{ncpu} threads x {reps} x 1 Mi-base chunks"}
    except Exception as e:  # the single-thread figure is the contract; this one is a bonus
        one["all_cores"] = {"error": str(e)}
    return one


def other_configs(ctx, cap, stream, dev, reps=5):
    """Kernel rates of the other BASELINE.json configs (parity-test cases, not the headline): C3 shape
    per GPU, C4, C5 strict and skip.  Resident data, HIP events on the library's stream, median of reps."""
    res = cap.Result()
    out = {}

    def timed(fn):
        fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    def synth(seed, n_bases, bits, amb=0):
        nw = (n_bases * bits + 63) // 64
        b = torch.empty(nw + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, b.data_ptr()), "kmers_synth_dna")
        return b

    def entry(name, ms, n_bases, alg_bytes):
        out[name] = {"kernel_ms": round(ms, 4), "Gbases_per_s": round(n_bases / ms / 1e6, 1),
                     "GB_per_s": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBPS, 4)}

    golden = 0x9E3779B97F4A7C15
    with torch.cuda.stream(stream):
        # C3: CanonicalDNAMers{31} over 10 Gbase LongDNA{2} sharded 8 ways -> 1.25 Gbase per GPU, kmers only
        L, K = 1_250_000_000, 31
        buf = synth(golden ^ 3, L, 2)
        a = torch.empty(L, dtype=torch.int64, device=dev)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 2, 0)
        ms = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, cap.MEM_DEVICE, C.byref(res)))
        entry("C3 CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2} (one of 8 shards), 8.25 B/kmer", ms, L, 8.25 * (L - K + 1))
        # C4: FwDNAMers{63} + reverse_complement over 1 Gbase LongDNA{4}
        L, K = 1_000_000_000, 63
        buf = synth(golden ^ 4, L, 4)
        a = torch.empty(2 * L, dtype=torch.int64, device=dev)
        b = torch.empty(2 * L, dtype=torch.int64, device=dev)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), cap.MEM_DEVICE, C.byref(res)))
        entry("C4 FwDNAMers{63} + reverse_complement, 1 Gbase LongDNA{4}, 32.5 B/kmer", ms, L, 32.5 * (L - K + 1))
        del b
        # C5: SpacedDNAMers{21,3} over 1 Gbase LongDNA{4}: strict, and the skip variant with N at p = 0.04
        K, J = 21, 3
        n = (L - K) // J + 1
        buf = synth(golden ^ 5, L, 4)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, a.data_ptr(), cap.MEM_DEVICE, C.byref(res)))
        entry("C5 SpacedDNAMers{21,3} strict, 1 Gbase LongDNA{4}, 9.5 B/kmer", ms, L, 0.5 * L + 8.0 * n)
        amb = synth(golden ^ 5, L, 4, 2621)
        seqa = cap.Seq(amb.data_ptr(), L, 0, 0, 4, 0)
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        st = torch.empty(m, dtype=torch.int64, device=dev)
        ms = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, a.data_ptr(), st.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)))
        entry(f"C5 skip variant (UnambiguousDNAMers{{21}} on the stride-3 lattice, p(N)=0.04, {m} kept), count+scan+emit", ms, L, 1.0 * L + 16.0 * m)
        del amb, st
        # fused consumers over the clean 1 Gbase LongDNA{4} (nothing materialised per kmer: no HBM roofline,
        # reported as kernel time and Gbases/s)
        val = C.c_uint64()
        ms = timed(lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
        out["fused XOR-reduce of CanonicalDNAMers{31} (test/benchmark.jl:9-15)"] = {"ms": round(ms, 4), "Gbases_per_s": round(L / ms / 1e6, 1)}
        sk = np.zeros(1000, dtype=np.uint64)
        ms = timed(lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)))
        out["fused MinHash sketch(fx_hash, CanonicalDNAMers{16}, 1000) (docs/src/minhash.md:34)"] = {"ms": round(ms, 4), "Gbases_per_s": round(L / ms / 1e6, 1)}
        counts = torch.empty(4 ** 8 // 2, dtype=torch.int64, device=dev)
        for Kc in (4, 8):
            ms = timed(lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), Kc, counts.data_ptr(), cap.MEM_DEVICE, C.byref(res)))
            out[f"fused composition counts of FwDNAMers{{{Kc}}} (docs/src/composition.md:28-39)"] = {"ms": round(ms, 4), "Gbases_per_s": round(L / ms / 1e6, 1)}
        # ragged batch: 8 M reads x 125 bases = the same 1 Gbase pool, CanonicalDNAMers{31} + fx_hash per read
        n_reads, rl, Kb = 8_000_000, 125, 31
        spans = torch.stack([torch.arange(n_reads, dtype=torch.int64, device=dev) * rl,
                             torch.full((n_reads,), rl, dtype=torch.int64, device=dev)], dim=1).contiguous()
        total = n_reads * (rl - Kb + 1)
        b = torch.empty(total, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, Kb, 2, a.data_ptr(),
                                               b.data_ptr(), 0, None, total, cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res)))
        out[f"kmers_batch: {n_reads} reads x {rl} bases, CanonicalDNAMers{{31}} + fx_hash per read"] = {
            "ms": round(ms, 4), "G_elements_per_s": round(total / ms / 1e6, 1), "Gbases_per_s": round(n_reads * rl / ms / 1e6, 1),
            "GB_per_s": round((16.0 * total + 0.5 * n_reads * rl) / ms / 1e6, 1)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=1_000_000_000, help="symbols per GPU (weak scaling)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--src-bits", type=int, default=4)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--max-grid", type=int, default=0)
    ap.add_argument("--no-hash", action="store_true", help="materialise canonical kmers only (8.5 / 8.25 B per kmer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra C3/C4/C5 rates (N = 1 only)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the k-mer kernels have no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev  # one rank per GPU; wraps only in the 1-GPU debug mode below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  KMERS_BENCH_BACKEND=gloo exists only to exercise the multi-rank
        # logic on a 1-GPU box (ranks share the device; not a measurement).
        backend = os.environ.get("KMERS_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import kmers_jl_amd as km
    from kmers_jl_amd.shard import HaloExchanger, plan_shards
    from oracle import pyoracle
    if rank == 0:
        pyoracle.build()  # the checker used after the timed region; one builder, the others wait
    if world > 1:
        dist.barrier()
    cap = km._capi
    ctx = km.Context(dev_index)
    stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
    if args.tile:
        ctx.set_param(cap.PARAM_TILE_KMERS, args.tile)
    if args.max_grid:
        ctx.set_param(cap.PARAM_MAX_GRID, args.max_grid)

    K, bits = args.k, args.src_bits
    total_bases = args.bases * world
    plan = plan_shards(total_bases, K, world, bits)
    sh = plan[rank]
    seed = 0x9E3779B97F4A7C15 ^ 2  # golden ^ config id (C2)
    N = cap.load().kmers_words_per_kmer(K, 2)

    with torch.cuda.stream(stream):
        buf = torch.zeros(sh.n_own_words + sh.halo_words + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words, bits, 0, buf.data_ptr()),
                  "kmers_synth_dna")
        out_k = torch.empty(sh.n_kmers * N, dtype=torch.int64, device=dev)
        out_h = None if args.no_hash else torch.empty(sh.n_kmers, dtype=torch.int64, device=dev)
        halo = HaloExchanger(buf, sh, plan)  # its workspace is filled on this stream too
    torch.cuda.synchronize()
    seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, bits, 0)
    res = cap.Result()
    flags = cap.MEM_DEVICE | cap.ASYNC
    ph = out_h.data_ptr() if out_h is not None else None

    def step(ev=None):
        with torch.cuda.stream(stream):
            if world > 1:
                halo.exchange()
            if ev:
                ev[0].record(stream)
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, out_k.data_ptr(), ph, 0, flags, C.byref(res))
            if ev:
                ev[1].record(stream)
        if rc != 0:
            raise RuntimeError(f"kmers_canonical failed: {ctx.last_error()}")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    rc, sres = ctx.sync()
    assert rc == 0, ctx.last_error()
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(events[i])
    fence()
    elapsed = time.perf_counter() - t0
    rc, sres = ctx.sync()
    assert rc == 0, ctx.last_error()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in events]))

    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms_max = float(t[0]), float(t[1])

    # ---- integrity of what the timed kernel wrote (outside the timed region) --------------
    verified = True
    with torch.cuda.stream(stream):
        # chunked so that the 10 Gbase size (165 GB of output) needs no large temporaries
        CH = 1 << 28
        col0 = out_k.view(-1, N)[:, 0]
        cmul = torch.tensor(FX_CONSTANT, dtype=torch.int64, device=dev)
        folded = 0
        for lo in range(0, sh.n_kmers, CH):
            hi = min(sh.n_kmers, lo + CH)
            if out_h is not None and N == 1:
                verified &= bool(torch.equal(out_h[lo:hi], col0[lo:hi] * cmul))
            folded ^= xor_fold(col0[lo:hi].contiguous())
        xr = C.c_uint64()
        seq_sync = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, bits, 0)
        ctx.check(ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq_sync), K, 2, 1, C.byref(xr), cap.MEM_DEVICE,
                                           C.byref(res)), "kmers_reduce_xor")
        verified &= folded == xr.value
    # oracle spot check on the first and last 2 Mbase of this rank's shard
    from oracle import pyoracle
    orc = pyoracle.get()
    probe = min(sh.n_kmers, 1 << 21)
    if probe:
        per_word = 64 // bits
        for first in sorted({0, ((sh.n_kmers - probe) // per_word) * per_word}):
            nb = min(probe, sh.n_kmers - first) + K - 1
            w = orc.synth_words(seed, sh.first_word + first // per_word, (nb * bits + 63) // 64 + 1, bits)
            ek, eh, eres = orc.canonical(w, nb, bits, 2, K)
            got_k = out_k.view(-1, N)[first:first + len(ek)].cpu().numpy().view(np.uint64)
            verified &= bool(np.array_equal(got_k, ek))
            if out_h is not None:
                verified &= bool(np.array_equal(out_h[first:first + len(eh)].cpu().numpy().view(np.uint64), eh))
    v = torch.tensor([1 if verified else 0], device=dev)
    if world > 1:
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
    verified = bool(v.item())

    if rank == 0:
        n_kmers_rank = sh.n_kmers
        bytes_per_kmer = bits / 8 + 8 * N + (0 if args.no_hash else 8)
        achieved = bytes_per_kmer * n_kmers_rank / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                d = json.load(open(pmc))
                if d.get("bases") == args.bases and d.get("k") == K and d.get("src_bits") == bits:
                    traffic = d.get("traffic_bytes_per_launch")
            except Exception:
                pass
        line = {
            "metric": METRIC, "value": round(total_bases * args.steps / elapsed / 1e9, 3), "unit": "Gbases/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"CanonicalDNAMers{{{K}}} + fx_hash over {args.bases / 1e9:g} Gbase LongDNA{{{bits}}} "
                                   f"per GPU (BASELINE.json configs[1]), kmers and hashes materialised in HBM"
                                   if not args.no_hash else
                                   f"CanonicalDNAMers{{{K}}} over {args.bases / 1e9:g} Gbase LongDNA{{{bits}}} per GPU",
                       "k": K, "src_bits": bits, "bases_per_gpu": args.bases,
                       "sharding": f"contiguous kmer-start ranges, (K-1)-base halo from rank+1 over RCCL each step ({halo.transport})"
                                   if world > 1 else "single shard",
                       "seed": hex(seed)},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
                         "kernel": "stream_kernel<src_bits,N,CANON,stride1>", "kernel_ms": round(kern_ms, 4),
                         "bytes_per_kmer": bytes_per_kmer, "kmers_per_launch": n_kmers_rank},
            "verified": verified,
        }
        if world == 1:
            try:  # what this device writes when it does nothing else: torch fills of the two output arrays
                with torch.cuda.stream(stream):
                    fills = []
                    for _ in range(5):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record(stream)
                        out_k.fill_(1)
                        if out_h is not None:
                            out_h.fill_(2)
                        e1.record(stream)
                        torch.cuda.synchronize()
                        fills.append(e0.elapsed_time(e1))
                    fill_bytes = out_k.numel() * 8 + (out_h.numel() * 8 if out_h is not None else 0)
                    fill_gbps = fill_bytes / (float(np.median(fills[1:])) * 1e-3) / 1e9
                line["roofline"]["torch_fill_GBps"] = round(fill_gbps, 1)  # torch.Tensor.fill_ over the same two arrays, same run
                line["roofline"]["vs_torch_fill"] = round(achieved / fill_gbps, 4)
            except Exception as e:
                line["roofline"]["torch_fill_GBps"] = None
                log(f"fill measurement failed: {e!r}")
        if world == 1 and not args.no_other_configs:
            try:  # informative extras; never allowed to break the headline line
                del out_k, out_h, buf
                torch.cuda.empty_cache()
                line["other_configs"] = other_configs(ctx, cap, stream, dev)
            except Exception as e:
                line["other_configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, bits, seed, args.bases, args.cpu_budget)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        raise SystemExit("bench output failed verification")


if __name__ == "__main__":
    main()
