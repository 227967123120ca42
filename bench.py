#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer iteration hot path on MI355X.

Workload (BASELINE.json configs[1]): CanonicalDNAMers{31} + fx_hash over 1 Gbase of synthetic
uniform LongDNA{4} per GPU, materialising both the canonical kmers and their hashes
(16.5 algorithmic bytes per kmer: 0.5 read + 8 + 8 written).  A "step" is one pass of the hot
path over the rank's shard: the (K-1)-base halo exchange with the next rank (N > 1 only: the C
ABI's kmers_halo_exchange on RCCL) followed by the canonical+hash kernel, inputs and outputs
resident in HBM.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before any HIP call
it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
...  bench.py <same arguments>` as a child, relays rank 0's JSON line and exits with the child's
status.  Started under torch.distributed.run (WORLD_SIZE set) it is one rank of the job.

Rank 0 prints ONE JSON line.  Weak scaling: every rank owns `--bases` symbols of one long
sequence of N * bases symbols; value = all symbols processed / max-over-ranks time.
"""
import argparse
import ctypes as C
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "canonical k-mers/sec (Gbases/s input) + % HBM roofline, K=31 DNA{4}"
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured float4 copy
FX_CONSTANT = 0x517CC1B727220A95
GOLDEN = 0x9E3779B97F4A7C15


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
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
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra C3/C4/C5/10 Gbase rates (N = 1 only)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic with rocprofv3 --pmc child runs")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--no-defrag", action="store_true", help="skip the one large allocate-and-release before the working buffers")
    ap.add_argument("--wake-s", type=float, default=1.0, help="seconds of plain fills before the W warm-up steps (a fresh or idle device is slower at first)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # internal: the profiled child of the traffic leg
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with N > 1 starts N ranks itself
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """Parent of an N-rank run.  Touches no GPU (torch.cuda.device_count() does not initialise HIP):
    the ranks are fresh child processes of torch.distributed.run."""
    import torch
    backend = os.environ.get("KMERS_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a HIP device: the k-mer kernels have no CPU fallback")
    if backend == "nccl" and ndev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {ndev} GPU(s) visible; RCCL needs one GPU per rank "
                         "(KMERS_BENCH_BACKEND=gloo lets ranks share a device to exercise the logic; not a measurement)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench.py launcher:", " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in p.stdout.splitlines():
        if not l.startswith("{"):
            log(l)
    if p.returncode != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank job failed with status {p.returncode}")
    if len(lines) != 1:
        raise SystemExit(f"bench.py: expected one JSON line from rank 0, got {len(lines)}")
    d = json.loads(lines[0])
    if d.get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the job reports n_gpus = {d.get('n_gpus')}")
    print(lines[0], flush=True)


# --------------------------------------------------------------------------------------------
def xor_fold(t):
    """XOR of all elements of an int64 CUDA tensor."""
    import torch
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
    import numpy as np

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


def verify_canonical(ctx, cap, stream, dev, buf, n_bases, first_kmer, first_word, bits, K, N, seed, out_k, out_h, n_kmers):
    """Integrity of what a canonical(+hash) launch wrote, over ALL elements: hashes == kmers * FX_CONSTANT (one-word kmers),
    XOR fold of the kmers == the fused reducer over the same sequence, and the first and last 2 Mbase against the oracle.
    Chunked so that the 10 Gbase size (165 GB of output) needs no large temporaries."""
    import numpy as np
    import torch

    from oracle import pyoracle
    ok = True
    res = cap.Result()
    with torch.cuda.stream(stream):
        CH = 1 << 28
        col0 = out_k.view(-1, N)[:, 0]
        cmul = torch.tensor(FX_CONSTANT, dtype=torch.int64, device=dev)
        folded = 0
        for lo in range(0, n_kmers, CH):
            hi = min(n_kmers, lo + CH)
            if out_h is not None and N == 1:
                ok &= bool(torch.equal(out_h[lo:hi], col0[lo:hi] * cmul))
            folded ^= xor_fold(col0[lo:hi].contiguous())
        xr = C.c_uint64()
        seq_sync = cap.Seq(buf.data_ptr(), n_bases, 0, first_kmer, bits, 0)
        ctx.check(ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq_sync), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)),
                  "kmers_reduce_xor")
        ok &= folded == xr.value
    orc = pyoracle.get()
    probe = min(n_kmers, 1 << 21)
    if probe:
        per_word = 64 // bits
        for first in sorted({0, ((n_kmers - probe) // per_word) * per_word}):
            nb = min(probe, n_kmers - first) + K - 1
            w = orc.synth_words(seed, first_word + first // per_word, (nb * bits + 63) // 64 + 1, bits)
            ek, eh, eres = orc.canonical(w, nb, bits, 2, K)
            got_k = out_k.view(-1, N)[first:first + len(ek)].cpu().numpy().view(np.uint64)
            ok &= bool(np.array_equal(got_k, ek))
            if out_h is not None:
                ok &= bool(np.array_equal(out_h[first:first + len(eh)].cpu().numpy().view(np.uint64), eh))
    return ok


def other_configs(ctx, cap, stream, dev, reps=7):
    """Kernel rates of the other BASELINE.json configs (parity-test cases, not the headline): C3 shape
    per GPU, C4, C5 strict and skip, and the north-star size (10 Gbase LongDNA{4}).  Resident data, HIP events on
    the library's stream, median of reps.  The 10 Gbase leg comes last: 165 GB of output per launch leave the device in a
    lower power state for a while (the 1.6 ms C3 launch measured 1.84 ms right behind it, profiles/r02_tuning.md)."""
    import numpy as np
    import torch
    res = cap.Result()
    out = {}

    def timed(fn):
        """HIP events on the library's stream around one call, median of reps.  Entry points with an asynchronous form
        (KMERS_ASYNC: kmers_canonical / kmers_fw / kmers_spaced) are timed in it, like the headline step: the events then
        bracket the kernel alone; around a synchronous call they would also bracket the host's wake-up after its stream
        wait and its next launch (measured: +0.25 ms on a 1.6 ms kernel, tools/diag_c3.py, profiles/r02_tuning.md).
        The device is kept busy with the same call for 50 ms first and the reps follow back to back: after an idle gap of
        0.2 s or more this device runs its next ~10 ms slower (the 1.59 ms C3 launch: 1.65, 1.87, 2.04, 1.95, 1.89, 1.79 ms
        in a row from a rested device, 1.59-1.61 right behind load; tools/diag_cooldown.py) -- a leg of a few launches
        would otherwise measure that transient, not the kernel."""
        fn()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.05:
            fn()
            fn()
            torch.cuda.synchronize()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            fn()
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        rc, _ = ctx.sync()
        assert rc == 0, ctx.last_error()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    ASYNC = cap.MEM_DEVICE | cap.ASYNC

    def synth(seed, n_bases, bits, amb=0):
        nw = (n_bases * bits + 63) // 64
        b = torch.empty(nw + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, b.data_ptr()), "kmers_synth_dna")
        return b

    def entry(name, ms, n_bases, alg_bytes, **extra):
        out[name] = {"kernel_ms": round(ms, 4), "Gbases_per_s": round(n_bases / ms / 1e6, 1),
                     "GB_per_s": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBPS, 4), **extra}

    with torch.cuda.stream(stream):
        # C3: CanonicalDNAMers{31} over 10 Gbase LongDNA{2} sharded 8 ways -> 1.25 Gbase per GPU, kmers only
        L, K = 1_250_000_000, 31
        buf = synth(GOLDEN ^ 3, L, 2)
        a = torch.empty(L, dtype=torch.int64, device=dev)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 2, 0)
        ms = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), None, 0, ASYNC, C.byref(res)))
        entry("C3 CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2} (one of 8 shards), 8.25 B/kmer", ms, L, 8.25 * (L - K + 1))
        # C4: FwDNAMers{63} + reverse_complement over 1 Gbase LongDNA{4}
        L, K = 1_000_000_000, 63
        buf = synth(GOLDEN ^ 4, L, 4)
        a = torch.empty(2 * L, dtype=torch.int64, device=dev)
        b = torch.empty(2 * L, dtype=torch.int64, device=dev)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), ASYNC, C.byref(res)))
        entry("C4 FwDNAMers{63} + reverse_complement, 1 Gbase LongDNA{4}, 32.5 B/kmer", ms, L, 32.5 * (L - K + 1))
        del b
        # C5: SpacedDNAMers{21,3} over 1 Gbase LongDNA{4}: strict, and the skip variant with N at p = 0.04
        K, J = 21, 3
        n = (L - K) // J + 1
        buf = synth(GOLDEN ^ 5, L, 4)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, a.data_ptr(), ASYNC, C.byref(res)))
        entry("C5 SpacedDNAMers{21,3} strict, 1 Gbase LongDNA{4}, 9.5 B/kmer", ms, L, 0.5 * L + 8.0 * n)
        amb = synth(GOLDEN ^ 5, L, 4, 2621)
        seqa = cap.Seq(amb.data_ptr(), L, 0, 0, 4, 0)
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        st = torch.empty(m, dtype=torch.int64, device=dev)
        ms = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, a.data_ptr(), st.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)))
        # algorithmic bytes: the source once (0.5 B/base) + (kmer, start) per kept element
        entry(f"C5 skip variant (UnambiguousDNAMers{{21}} on the stride-3 lattice, p(N)=0.04, {m} kept), 0.5 B/base + 16 B/kept", ms, L, 0.5 * L + 16.0 * m)
        # the reference's own skipping iterator at the headline K: UnambiguousDNAMers{31}, same source
        K = 31
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        st = torch.empty(m, dtype=torch.int64, device=dev)
        ms = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, 1, a.data_ptr(), st.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)))
        entry(f"UnambiguousDNAMers{{31}}, 1 Gbase LongDNA{{4}}, p(N)=0.04, {m} kept, 0.5 B/base + 16 B/kept", ms, L, 0.5 * L + 16.0 * m)
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
        del a, b, buf, spans, counts
        torch.cuda.empty_cache()
        # N1 (north star): CanonicalDNAMers{31} + fx_hash over 10 Gbase LongDNA{4} on ONE GPU: 5 GB in, 160 GB out
        free_b, _ = torch.cuda.mem_get_info(dev)
        L, K = 10_000_000_000, 31
        if free_b > 175e9:
            seed10 = GOLDEN ^ 10
            buf = synth(seed10, L, 4)
            n = L - K + 1
            a = torch.empty(n, dtype=torch.int64, device=dev)
            h = torch.empty(n, dtype=torch.int64, device=dev)
            seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
            ms = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), h.data_ptr(), 0, ASYNC, C.byref(res)))
            ok = verify_canonical(ctx, cap, stream, dev, buf, L, 0, 0, 4, K, 1, seed10, a, h, n)
            entry("N1 north star: CanonicalDNAMers{31} + fx_hash, 10 Gbase LongDNA{4}, one GPU, 16.5 B/kmer", ms, L, 16.5 * n, verified=ok)
            del buf, a, h
        else:
            out["N1 north star: 10 Gbase LongDNA{4}"] = {"skipped": f"needs 165 GB of HBM, {free_b / 1e9:.0f} GB free"}
    return out


# --------------------------------------------------------------------------------------------
# roofline.traffic: HBM bytes of the headline kernel from the PMC counters, measured in THIS run by two profiled child
# processes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots")
def pmc_child(args):
    """The profiled program: the headline launch three times, nothing else (run under rocprofv3 --pmc)."""
    import torch

    import kmers_jl_amd as km
    cap = km._capi
    ctx = km.Context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
    K, bits, L = args.k, args.src_bits, args.bases
    nw = (L * bits + 63) // 64
    with torch.cuda.stream(stream):
        buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, GOLDEN ^ 2, 0, nw, bits, 0, buf.data_ptr()), "kmers_synth_dna")
        n = L - K + 1
        N = cap.load().kmers_words_per_kmer(K, 2)
        out_k = torch.empty(n * N, dtype=torch.int64, device=dev)
        out_h = None if args.no_hash else torch.empty(n, dtype=torch.int64, device=dev)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    for _ in range(3):
        rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, out_k.data_ptr(), out_h.data_ptr() if out_h is not None else None,
                                     0, cap.MEM_DEVICE, C.byref(res))
        assert rc == 0, ctx.last_error()
    torch.cuda.synchronize()


def measure_traffic(args):
    """Returns (bytes per launch or None, description of where the number comes from)."""
    import csv
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="kmers_pmc_")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--pmc-child", "--bases", str(args.bases), "--k", str(args.k), "--src-bits", str(args.src_bits)]
            if args.no_hash:
                cmd.append("--no-hash")
            env = dict(os.environ, TMPDIR=tmp)
            p = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if p.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child failed ({p.returncode}): {p.stderr[-300:]}"
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "stream_kernel" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        rows.append(float(r["Counter_Value"]))
            if not rows:
                return None, f"no {counter} rows for stream_kernel in the rocprofv3 output"
            vals[counter] = sum(rows) / len(rows)
    except Exception as e:
        return None, f"PMC pass failed: {e!r}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # units and gfx950 corrections exactly as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB;
    # FETCH_SIZE reports half of a coalesced streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16 B/lane stores
    traffic = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return traffic, (f"measured in this run: two rocprofv3 --pmc child passes over the same launch (FETCH_SIZE {vals['FETCH_SIZE']:.1f} KiB "
                     f"x 2 [gfx950 correction], WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB)")


def replayed_traffic(args):
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        d = json.load(open(pmc))
        if d.get("bases") == args.bases and d.get("k") == args.k and d.get("src_bits") == args.src_bits and not args.no_hash:
            return d.get("traffic_bytes_per_launch"), "REPLAYED from profiles/pmc_traffic.json (an earlier collection, not this run)"
    except Exception:
        pass
    return None, "not measured"


# --------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={env_world}: launch exactly --gpus ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the k-mer kernels have no CPU fallback")
    ndev = torch.cuda.device_count()
    backend = "none"
    # KMERS_BENCH_FORCE_GROUP=1 (tests): also a 1-rank run goes through the N > 1 code path -- process group, ncclUniqueId
    # hand-over, kmers_comm_create, kmers_halo_exchange every step, the two reductions -- which is all of it that a 1-GPU
    # box can run on RCCL
    grouped = env_world > 1 or (os.environ.get("KMERS_BENCH_FORCE_GROUP") == "1" and "RANK" in os.environ)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm.  KMERS_BENCH_BACKEND=gloo exists only to exercise the multi-rank
        # logic on a 1-GPU box (ranks share the device; not a measurement).
        backend = os.environ.get("KMERS_BENCH_BACKEND", "nccl")
        if backend == "nccl" and ndev < env_world:
            raise SystemExit(f"{env_world} RCCL ranks need {env_world} GPUs, {ndev} visible")
    dev_index = local_rank % ndev  # one rank per GPU; wraps only in the shared-device gloo mode
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if grouped:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        backend = dist.get_backend()
    world = dist.get_world_size() if grouped else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the process group has {world} ranks")

    import kmers_jl_amd as km
    from kmers_jl_amd.shard import HaloExchanger, NativeComm, plan_shards
    from oracle import pyoracle
    if rank == 0:
        pyoracle.build()  # the checker used after the timed region; one builder, the others wait
    if grouped:
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
    seed = GOLDEN ^ 2  # golden ^ config id (C2)
    N = cap.load().kmers_words_per_kmer(K, 2)

    # the halo transport: under RCCL the C ABI's own exchange (grouped ncclSend/ncclRecv on the context's stream; torch only
    # hands the 128-byte ncclUniqueId around); under gloo (shared-device debugging, CPU tests) torch.distributed's
    transport = os.environ.get("KMERS_HALO_TRANSPORT", "native" if backend == "nccl" else "allgather")
    comm = None
    # A fresh box hands out fragmented VRAM: the same launch over two freshly allocated 8 GB output arrays runs at 0.794 of
    # 8 TB/s in the first process of a box and at 0.83-0.84 in any process after one that has allocated and released a large
    # block (tools/diag_alloc2.py, profiles/r02_tuning.md section 7: the driver then has large contiguous ranges to give, and
    # the address translation reaches further).  One allocation of 60 % of the free memory, released at once and never
    # touched, puts every run in the second state; INTEGRATION.md gives the same advice to host applications.
    defrag_gb = 0.0
    shared_device = grouped and backend != "nccl"  # (gloo debugging mode: the ranks share one device and its memory)
    if not args.no_defrag and not shared_device:
        free_b, _total_b = torch.cuda.mem_get_info(dev)
        n_defrag = int(free_b * 0.6) // 8
        try:
            tmp = torch.empty(n_defrag, dtype=torch.int64, device=dev)
            del tmp
            defrag_gb = round(n_defrag * 8 / 1e9, 1)
        except torch.OutOfMemoryError:
            pass  # somebody else holds the memory: measure in whatever state the allocator is
        torch.cuda.empty_cache()
    with torch.cuda.stream(stream):
        buf = torch.zeros(sh.n_own_words + sh.halo_words + 2, dtype=torch.int64, device=dev)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words, bits, 0, buf.data_ptr()),
                  "kmers_synth_dna")
        out_k = torch.empty(sh.n_kmers * N, dtype=torch.int64, device=dev)
        out_h = None if args.no_hash else torch.empty(sh.n_kmers, dtype=torch.int64, device=dev)
        halo = None
        if grouped and transport != "native":
            halo = HaloExchanger(buf, sh, plan, transport=transport)  # its workspace is filled on this stream too
    if grouped and transport == "native":
        if backend != "nccl":
            raise SystemExit("KMERS_HALO_TRANSPORT=native needs one GPU per rank (RCCL); the gloo mode shares a device")
        comm = NativeComm.bootstrap(ctx)
        shard_c = comm._shard_struct(sh)
    torch.cuda.synchronize()
    seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, bits, 0)
    res = cap.Result()
    flags = cap.MEM_DEVICE | cap.ASYNC
    ph = out_h.data_ptr() if out_h is not None else None

    def step(ev=None):
        with torch.cuda.stream(stream):
            if comm is not None:
                comm.halo_exchange(shard_c, buf.data_ptr())
            elif halo is not None:
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
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    # Wake the device: after an idle gap (allocation, data generation, process start) this device runs its next ~10 ms
    # 10-25 % slower (tools/diag_cooldown.py, profiles/r02_tuning.md section 1) and the first second of a fresh process 2 %
    # slower than the ninety that follow (tools/diag_warmup.py: 2.49 ms per launch, then 2.435), which is longer than the W
    # warm-up steps of the contract.  --wake-s seconds (default 1) of plain torch fills of the output arrays -- not steps of
    # the hot path -- come first; the W warm-up steps and the K timed steps below are exactly the contract's.
    with torch.cuda.stream(stream):
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < args.wake_s:
            out_k.fill_(0)
            if out_h is not None:
                out_h.fill_(0)
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
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms_max = float(t[0]), float(t[1])

    # ---- integrity of what the timed kernel wrote (outside the timed region) --------------
    verified = verify_canonical(ctx, cap, stream, dev, buf, sh.n_bases, sh.first_kmer, sh.first_word, bits, K, N, seed, out_k, out_h, sh.n_kmers)
    # the two tiny cross-shard reductions of the path, through the same communicator (results known in closed form)
    if comm is not None:
        st, pos, enc = comm.first_error(1 if rank == world - 1 else 0, err_pos=sh.first_base + 5, err_enc=0xF)
        verified &= (st, pos, enc) == (1, plan[-1].first_base + 5, 0xF)
        off, tot = comm.output_offsets(sh.n_kmers)
        verified &= off == sh.first_kmer and tot == sum(s.n_kmers for s in plan)
    v = torch.tensor([1 if verified else 0], device=dev)
    if grouped:
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
    verified = bool(v.item())

    if rank == 0:
        n_kmers_rank = sh.n_kmers
        bytes_per_kmer = bits / 8 + 8 * N + (0 if args.no_hash else 8)
        achieved = bytes_per_kmer * n_kmers_rank / (kern_ms * 1e-3) / 1e9
        if not grouped:
            sharding = "single shard"
        else:
            how = {"native": "kmers_halo_exchange of the C ABI: grouped ncclSend/ncclRecv on the kernel's stream (RCCL; torch carried only the ncclUniqueId)",
                   "allgather": f"torch.distributed all_gather of <= 32 B per rank ({backend})",
                   "p2p": f"torch.distributed batch_isend_irecv between neighbours ({backend})"}[transport]
            sharding = f"contiguous kmer-start ranges, (K-1)-base halo from rank+1 each step; backend {backend}; transport {transport}: {how}"
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
                       "sharding": sharding, "backend": backend, "halo_transport": transport if grouped else None,
                       "seed": hex(seed), "wake_s": args.wake_s,
                       "vram_defrag_GB": defrag_gb},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": "not measured",
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
        if world == 1:
            del out_k, out_h, buf
            torch.cuda.empty_cache()
            traffic, source = (None, "not measured (--no-pmc)") if args.no_pmc else measure_traffic(args)
            if traffic is None:
                log(f"roofline.traffic: {source}")
                traffic, source2 = replayed_traffic(args)
                source = f"{source2}; live measurement: {source}"
            line["roofline"]["traffic"] = traffic
            line["roofline"]["traffic_source"] = source
        if world == 1 and not args.no_other_configs:
            try:  # informative extras; never allowed to break the headline line
                line["other_configs"] = other_configs(ctx, cap, stream, dev)
            except Exception as e:
                line["other_configs"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(K, bits, seed, args.bases, args.cpu_budget)
        print(json.dumps(line), flush=True)
    if comm is not None:
        comm.close()
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    if not verified:
        raise SystemExit("bench output failed verification")


if __name__ == "__main__":
    main()
