#!/usr/bin/env python3
"""bench.py -- headline benchmark of the k-mer iteration hot path on MI355X.

Workload (BASELINE.json configs[1]): CanonicalDNAMers{31} + fx_hash over 1 Gbase of synthetic
uniform LongDNA{4} per GPU, materialising both the canonical kmers and their hashes
(16.5 algorithmic bytes per kmer: 0.5 read + 8 + 8 written).  A "step" is one pass of the hot
path over the rank's shard: the (K-1)-base halo exchange with the next rank (N > 1 only: the C
ABI's kmers_halo_exchange on RCCL) followed by the canonical+hash kernel, inputs and outputs
resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--total-bases T]

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: before any HIP call
it starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
...  bench.py <same arguments>` as a child, relays rank 0's JSON line and exits with the child's
status.  Started under torch.distributed.run (WORLD_SIZE set) it is one rank of the job.

Rank 0 prints ONE JSON line.  Two scaling modes:
  weak   (default)            every rank owns `--bases` symbols of one long sequence of N * bases symbols
  strong (--total-bases T)    ONE sequence of T symbols (the north star: 10 Gbase LongDNA{4}) is split over the N ranks
                              by kmers_shard_plan; `"scaling": "strong"`
value = all symbols processed / max-over-ranks time either way.  A weak run on N > 1 GPUs ALSO measures the strong split of
the north-star input (`strong_scaling` in the line: K steps of 10 Gbase over the N ranks under the same barrier and
max-over-ranks rule, then the same 10 Gbase on rank 0 alone, and the ratio of the two) so that the contract's plain
`--gpus N` command yields the fixed-input speedup as well.

Memory: the buffers come from the library's allocator (kmers_dev_alloc, include/kmers_hip.h) -- what a Julia or C host gets:
by default the device's CLASS POOL (1 GiB handles of HIP's virtual-memory management, each block assembled by HBM region
class; no reservation), with `--alloc plain` torch allocations; `roofline.plain_alloc` reports the same launch into plain
allocations made before the pool existed.  `fresh_outputs_per_step` is the same K steps with FRESH outputs per step --
{kmers_dev_alloc, kmers_dev_alloc, launch, kmers_dev_free, kmers_dev_free}, what `collect` per sequence amounts to (Base.collect
over src/iterators/CanonicalKmers.jl:199-225 makes a new Vector per call) -- beside the resident headline.
"""
import argparse
import ctypes as C
import datetime
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
NORTH_STAR_BASES = 10_000_000_000
N_SIMDS = 1024  # 256 CUs x 4 SIMDs
# What a SIMD of this device issues (tools/device_probes/valu_rates.hip, profiles/r04_valu_rates.txt; cycles per wave64 vector instruction,
# instructions of a launch / the launch's span, two or more wavefronts per SIMD): 2.24 for the simple two-operand integer
# instructions over registers or literals (v_and / v_or / v_xor / v_add / v_sub / v_mov / right shifts; v_bitop3 over three
# registers), 4.1 for everything else these kernels use (v_alignbit, v_perm, v_lshl_or, v_bfe, v_cndmask, compares, multiplies,
# every 64-bit shift, v_lshlrev_b32, anything with an SGPR operand) -- ONE wavefront alone already
# issues one every 4.5-4.9 cycles, i.e. more wavefronts buy nothing for that class.  (The first version of the tool divided by the
# mean LIFETIME of the wavefronts instead of the span; the arbiter prefers the oldest wavefront, lifetimes are staggered, and it
# reported 1.32 / 2.35.  The "4 cycles per instruction" of rounds 2-3 was right for the slow class.)  A true floor prices every
# instruction at the fast class.
VALU_CYCLES_FAST, VALU_CYCLES_SLOW = 2.24, 4.1


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bases", type=int, default=1_000_000_000, help="symbols per GPU (weak scaling, the default)")
    ap.add_argument("--total-bases", type=int, default=0, help="strong scaling: ONE sequence of this many symbols split over the --gpus ranks")
    ap.add_argument("--strong-bases", type=int, default=-1,
                    help="weak runs on N > 1 GPUs also time this fixed input split over the N ranks (default: the north star's 10 Gbase; 0 = skip)")
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--src-bits", type=int, default=4)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--max-grid", type=int, default=0)
    ap.add_argument("--no-hash", action="store_true", help="materialise canonical kmers only (8.5 / 8.25 B per kmer)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the extra C3/C4/C5/10 Gbase rates (N = 1 only)")
    ap.add_argument("--no-pmc", action="store_true", help="do not measure roofline.traffic / VALU issue shares with rocprofv3 --pmc child runs")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--alloc", choices=("pool", "plain"), default="pool",
                    help="pool: buffers from kmers_dev_alloc, served by the device's class pool (the product's default); plain: torch allocations")
    ap.add_argument("--wake-s", type=float, default=1.0, help="seconds of plain fills before the W warm-up steps (a fresh or idle device is slower at first)")
    ap.add_argument("--pmc-child", default="", help=argparse.SUPPRESS)  # internal: the profiled child ("headline" or "legs")
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
# device memory: the library's blocks, viewed as torch tensors for the checks
class _RawDeviceArray:
    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (n_words,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None}


class Memory:
    """Where the bench's buffers come from.  use_lib: ctx.alloc (kmers_dev_alloc: the device's class pool) wrapped as int64 torch
    tensors through __cuda_array_interface__; else torch.empty."""

    def __init__(self, ctx, dev, use_lib):
        self.ctx, self.dev, self.use_lib, self.live = ctx, dev, use_lib, {}

    def room(self):
        """bytes a new block can still get"""
        import torch
        free = torch.cuda.mem_get_info(self.dev)[0]
        if self.use_lib:
            st = self.ctx.pool_stats()
            free += st["cached"] + st["free"]  # (the pool takes its own idle memory before it asks the driver)
        return free

    def empty(self, n_words, lone_output=False):
        """lone_output: the only output array of the launches that fill it (kmers_dev_alloc_role(KMERS_ALLOC_LONE_OUTPUT))."""
        import torch
        n_words = max(int(n_words), 1)
        if not self.use_lib:
            return torch.empty(n_words, dtype=torch.int64, device=self.dev)
        ptr = self.ctx.alloc(8 * n_words, lone_output=lone_output)
        t = torch.as_tensor(_RawDeviceArray(ptr, n_words), device=self.dev)
        assert t.data_ptr() == ptr and t.numel() == n_words
        self.live[ptr] = t
        return t

    def free(self, *tensors):
        """Give the blocks back to the library (the tensors must not be used afterwards).  kmers_dev_free orders a block's next
        use behind the work queued on the LIBRARY's streams; torch may have touched these tensors on streams of its own (the
        checks): the device is waited for first, as include/kmers_hip.h asks of a host framework."""
        import torch
        if self.use_lib and any(t is not None for t in tensors):
            torch.cuda.synchronize()
        for t in tensors:
            if t is None:
                continue
            if self.use_lib and t.data_ptr() in self.live:
                ptr = t.data_ptr()
                del self.live[ptr]
                self.ctx.free(ptr)


def run_lengths(seq):
    out = []
    for c in seq:
        if out and out[-1][0] == c:
            out[-1][1] += 1
        else:
            out.append([c, 1])
    return out


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
    """Integrity of what a canonical(+hash) launch wrote, over ALL elements.  Two kinds of check:
      * against the ORACLE: the first and the last 2 Mbase of the shard, kmers and hashes, bit for bit (the independent part);
      * SELF-CONSISTENCY of the HIP path over everything in between: hashes == kmers * FX_CONSTANT (one-word kmers; torch
        arithmetic against the kernel's), and the XOR fold of the kmers == kmers_reduce_xor over the same sequence -- a
        different kernel of the same library (run_kernel.hpp), so HIP against HIP: it catches a launch that skipped or
        misplaced elements, not an error both kernels share.
    Chunked so that the 10 Gbase size (165 GB of output) needs no large temporaries."""
    import numpy as np
    import torch

    from oracle import pyoracle
    ok = True
    res = cap.Result()
    if n_kmers == 0:
        return True
    with torch.cuda.stream(stream):
        CH = 1 << 28
        col0 = out_k.view(-1, N)[:n_kmers, 0]
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


def busy_timed(ctx, stream, fn, reps=7, busy_s=0.05):
    """HIP events on the library's stream around one call, median of reps.  Entry points with an asynchronous form
    (KMERS_ASYNC: kmers_canonical / kmers_fw / kmers_spaced) are timed in it, like the headline step: the events then
    bracket the kernel alone; around a synchronous call they would also bracket the host's wake-up after its stream
    wait and its next launch (measured: +0.25 ms on a 1.6 ms kernel, profiles/r02_tuning.md section 1).
    The device is kept busy with the same call for 50 ms first and the reps follow back to back: after an idle gap of
    0.2 s or more this device runs its next ~10 ms slower -- a leg of a few launches would otherwise measure that
    transient, not the kernel."""
    import numpy as np
    import torch
    fn()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < busy_s:
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


def other_configs(ctx, cap, stream, dev, mem, valu=None, reps=7, write_ceiling_gbps=HBM_PEAK_GBPS, write_ceiling_source="8 TB/s spec (no pool)"):
    """Kernel rates of the other BASELINE.json configs (parity-test cases, not the headline): C3 shape
    per GPU, C4, C5 strict and skip, and the north-star size (10 Gbase LongDNA{4}).  Resident data, HIP events on
    the library's stream, median of reps.  The 10 Gbase leg comes last: 165 GB of output per launch leave the device in a
    lower power state for a while (the 1.6 ms C3 launch measured 1.84 ms right behind it, profiles/r02_tuning.md).
    `valu`: per-leg VALU issue shares from the PMC child (measure_legs), merged into the entries they belong to."""
    import numpy as np
    import torch
    res = cap.Result()
    out = {}
    valu = valu or {}

    def timed(fn):
        return busy_timed(ctx, stream, fn, reps)

    ASYNC = cap.MEM_DEVICE | cap.ASYNC

    def synth(seed, n_bases, bits, amb=0):
        nw = (n_bases * bits + 63) // 64
        b = mem.empty(nw + 2)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, b.data_ptr()), "kmers_synth_dna")
        return b

    def entry(name, ms, n_bases, alg_bytes, **extra):
        out[name] = {"kernel_ms": round(ms, 4), "Gbases_per_s": round(n_bases / ms / 1e6, 1),
                     "GB_per_s": round(alg_bytes / ms / 1e6, 1), "frac_of_8TBps": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBPS, 4), **extra}

    def ceilings(leg, ms, alg_bytes):
        """What bounds a leg that is not purely a store stream.  Every ratio is formed inside ONE profiled dispatch (its duration,
        its SQ_INSTS_VALU and its GRBM_GUI_ACTIVE, measure_legs); the unprofiled time of this run stands beside them, it is
        not divided into them.  hbm_floor: algorithmic bytes at the two-stream rate the pool's probes MEASURED for two region
        classes (kmers_pool_info; 8 TB/s spec without a pool); valu_floor: every vector instruction at the fastest rate
        the SIMDs issue (VALU_CYCLES_FAST) -- a floor no instruction mix can beat, `valu_floor_ms_all_slow` the same at the
        rate of every other instruction class; frac_of_max_floor = the larger floor / the profiled duration (<= 1 by construction)."""
        v = valu.get(leg)
        if not v:
            return {"valu_issue": "not measured (no PMC pass)"}
        hbm_floor = alg_bytes / (write_ceiling_gbps * 1e9) * 1e3 if alg_bytes else 0.0
        floor = max(hbm_floor, v["valu_floor_ms"] or 0.0)
        prof = v["profiled_ms"]
        return {"profiled_ms": prof, "profiled_clock_GHz": v["clock_GHz"], "unprofiled_ms_this_run": round(ms, 4),
                "valu_insts_per_launch": v["valu_insts"], "valu_cycles_per_inst_per_simd": v["cycles_per_inst"],
                "valu_floor_ms": v["valu_floor_ms"], "valu_floor_ms_all_slow": v["valu_floor_ms_slow"],
                "valu_frac_of_floor": round((v["valu_floor_ms"] or 0.0) / prof, 4) if prof else None,
                "hbm_floor_ms": round(hbm_floor, 4), "hbm_floor_rate_GBps": round(write_ceiling_gbps, 1), "hbm_floor_rate_source": write_ceiling_source,
                "frac_of_max_floor": round(floor / prof, 4) if prof else None, "valu_source": v["source"]}

    with torch.cuda.stream(stream):
        # C3: CanonicalDNAMers{31} over 10 Gbase LongDNA{2} sharded 8 ways -> 1.25 Gbase per GPU, kmers only
        L, K = 1_250_000_000, 31
        # (a launch with ONE output array takes it from the pool by role -- kmers_dev_alloc_role(KMERS_ALLOC_LONE_OUTPUT): across a
        # class boundary of HBM, written through two windows; sized to the launch, as a host's `collect` would)
        lone = mem.empty(L - K + 1, lone_output=True)
        buf = synth(GOLDEN ^ 3, L, 2)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 2, 0)
        ms = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, lone.data_ptr(), None, 0, ASYNC, C.byref(res)))
        entry("C3 CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2} (one of 8 shards), 8.25 B/kmer", ms, L, 8.25 * (L - K + 1))
        mem.free(buf, lone)
        a = mem.empty(2 * 1_000_000_000)
        b = mem.empty(2 * 1_000_000_000)
        log("[bench] other configs: C3 done")
        # C4: FwDNAMers{63} + reverse_complement over 1 Gbase LongDNA{4}
        L, K = 1_000_000_000, 63
        buf = synth(GOLDEN ^ 4, L, 4)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), b.data_ptr(), ASYNC, C.byref(res)))
        entry("C4 FwDNAMers{63} + reverse_complement, 1 Gbase LongDNA{4}, 32.5 B/kmer", ms, L, 32.5 * (L - K + 1))
        mem.free(buf)
        log("[bench] C4 done")
        # C5: SpacedDNAMers{21,3} over 1 Gbase LongDNA{4}: strict, and the skip variant with N at p = 0.04
        K, J = 21, 3
        n = (L - K) // J + 1
        lone = mem.empty(n, lone_output=True)
        buf = synth(GOLDEN ^ 5, L, 4)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
        ms = timed(lambda: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, lone.data_ptr(), ASYNC, C.byref(res)))
        entry("C5 SpacedDNAMers{21,3} strict, 1 Gbase LongDNA{4}, 9.5 B/kmer", ms, L, 0.5 * L + 8.0 * n)
        mem.free(lone)
        amb = synth(GOLDEN ^ 5, L, 4, 2621)
        seqa = cap.Seq(amb.data_ptr(), L, 0, 0, 4, 0)
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        ms = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, J, a.data_ptr(), b.data_ptr(), m, ASYNC, C.byref(res)))
        # algorithmic bytes: the source once (0.5 B/base) + (kmer, start) per kept element
        alg = 0.5 * L + 16.0 * m
        entry(f"C5 skip variant (UnambiguousDNAMers{{21}} on the stride-3 lattice, p(N)=0.04, {m} kept), 0.5 B/base + 16 B/kept", ms, L, alg,
              **ceilings("u21", ms, alg))
        # the reference's own skipping iterator at the headline K: UnambiguousDNAMers{31}, same source
        K = 31
        ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)), "count")
        m = int(res.n_out)
        ms = timed(lambda: ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), K, 1, a.data_ptr(), b.data_ptr(), m, ASYNC, C.byref(res)))
        alg = 0.5 * L + 16.0 * m
        entry(f"UnambiguousDNAMers{{31}}, 1 Gbase LongDNA{{4}}, p(N)=0.04, {m} kept, 0.5 B/base + 16 B/kept", ms, L, alg, **ceilings("u31", ms, alg))
        mem.free(amb)
        log("[bench] C5 / UnambiguousKmers done")
        # fused consumers over the clean 1 Gbase LongDNA{4} (nothing materialised per kmer: their roofline is the integer
        # issue rate, SURVEY.md 8d -- kmers/s and the VALU issue share, not HBM bytes)
        def fused(name, leg, ms, n_kmers):
            out[name] = {"ms": round(ms, 4), "Gbases_per_s": round(L / ms / 1e6, 1), "G_kmers_per_s": round(n_kmers / ms / 1e6, 1),
                         "bound": "instruction issue (nothing is materialised: 0.5 B/base of HBM reads); valu_cycles_per_inst_per_simd "
                                  "against the 2.24-4.1 the SIMDs can issue says how far from it", **ceilings(leg, ms, 0.0)}
        val = C.c_uint64()
        ms = timed(lambda: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)))
        fused("fused XOR-reduce of CanonicalDNAMers{31} (test/benchmark.jl:9-15)", "xor", ms, L - 30)
        sk = np.zeros(1000, dtype=np.uint64)
        ms = timed(lambda: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)))
        fused("fused MinHash sketch(fx_hash, CanonicalDNAMers{16}, 1000) (docs/src/minhash.md:34)", "minhash", ms, L - 15)
        for Kc in (4, 8):
            ms = timed(lambda: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), Kc, b.data_ptr(), cap.MEM_DEVICE, C.byref(res)))
            fused(f"fused composition counts of FwDNAMers{{{Kc}}} (docs/src/composition.md:28-39)", f"comp{Kc}", ms, L - Kc + 1)
        log("[bench] fused consumers done")
        # ragged batch: 8 M reads x 125 bases = the same 1 Gbase pool, CanonicalDNAMers{31} + fx_hash per read
        n_reads, rl, Kb = 8_000_000, 125, 31
        spans = torch.stack([torch.arange(n_reads, dtype=torch.int64, device=dev) * rl,
                             torch.full((n_reads,), rl, dtype=torch.int64, device=dev)], dim=1).contiguous()
        total = n_reads * (rl - Kb + 1)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, Kb, 2, a.data_ptr(),
                                               b.data_ptr(), 0, None, total, cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res)))
        out[f"kmers_batch: {n_reads} reads x {rl} bases, CanonicalDNAMers{{31}} + fx_hash per read"] = {
            "ms": round(ms, 4), "G_elements_per_s": round(total / ms / 1e6, 1), "Gbases_per_s": round(n_reads * rl / ms / 1e6, 1),
            "GB_per_s": round((16.0 * total + 0.5 * n_reads * rl) / ms / 1e6, 1),
            "frac_of_8TBps": round((16.0 * total + 0.5 * n_reads * rl) / ms / 1e6 / HBM_PEAK_GBPS, 4)}
        del spans
        log("[bench] kmers_batch (clean 4-bit pool) done")
        # SURVEY.md 8(f) rows, the same 1 Gbase: f1 the headline launch from TEXT (1 B/base in: String / Vector{UInt8} sources,
        # FwKmers.jl:117-129), f3 a 4-bit kmer alphabet (Copyable 4 -> 4, two-word kmers), f4 element-wise fx_hash and
        # reverse_complement over an array of kmers (kmer.jl:255-261, transformations.jl:32-34)
        Kf = 31
        idx = torch.randint(0, 4, (L,), dtype=torch.uint8, device=dev)
        text_words = mem.empty(L // 8 + 2)                                              # (the text comes from kmers_dev_alloc like every other buffer)
        text = text_words.view(torch.uint8)
        text.fill_(65)                                                                   # "A"
        for code, add in ((1, 2), (2, 6), (3, 19)):                                        # "C", "G", "T"
            text[:L] += idx.eq(code).to(torch.uint8) * add
        del idx
        tseq = cap.Seq(text.data_ptr(), L, 0, 0, 8, 0)
        torch.cuda.synchronize()
        ms = timed(lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(tseq), Kf, 2, a.data_ptr(), b.data_ptr(), 0, ASYNC, C.byref(res)))
        entry("f1 CanonicalDNAMers{31} + fx_hash from 1 Gbase of ASCII text (String source), 17 B/kmer", ms, L, 17.0 * (L - Kf + 1))
        log("[bench] f1 done")
        # kmers_batch on the reads people have (docs/src/minhash.md:31-35, docs/src/faq.md:28-33,
        # src/iterators/UnambiguousKmers.jl:109-132): the same 8 M x 125 from TEXT; with an N in 1 % / 10 % of the reads under
        # KMERS_BATCH_SKIP (4-bit pool and text); lengths uniform 50-250 (about 6.7 M reads of the same pool).  Whole calls.
        try:
            out.update(batch_read_legs(ctx, cap, dev, seq, tseq, buf, text, a, b, timed, L))
        except Exception as e:  # noqa: BLE001
            out["kmers_batch on realistic reads"] = {"error": repr(e)}
        log("[bench] kmers_batch on realistic reads done")
        del text
        mem.free(text_words)
        del text_words
        mem.free(a)
        lone = mem.empty(2 * (L - Kf + 1), lone_output=True)
        ms = timed(lambda: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), Kf, 4, lone.data_ptr(), None, ASYNC, C.byref(res)))
        entry("f3 FwKmers{DNAAlphabet{4},31} (4-bit kmer alphabet, two-word kmers), 1 Gbase LongDNA{4}, 16.5 B/kmer", ms, L, 16.5 * (L - Kf + 1))
        mem.free(lone)
        a = mem.empty(2 * 1_000_000_000)
        ms = timed(lambda: ctx.lib.kmers_fx_hash(ctx.handle, a.data_ptr(), 1, L, 0, b.data_ptr(), ASYNC))
        entry("f4 fx_hash over an array of 1 G one-word kmers, 16 B/kmer", ms, L, 16.0 * L)
        ms = timed(lambda: ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, a.data_ptr(), Kf, 2, L, b.data_ptr(), ASYNC))
        entry("f4 reverse_complement over an array of 1 G DNAKmer{31}, 16 B/kmer", ms, L, 16.0 * L)
        mem.free(a, b, buf)
        del a, b, buf
        torch.cuda.empty_cache()
        log("[bench] f3 / f4 done")
        try:
            out["e2e host pointers (H2D + kernel + D2H; never `value`)"] = host_pointer_path(ctx, cap, dev)
        except Exception as e:  # noqa: BLE001
            out["e2e host pointers (H2D + kernel + D2H; never `value`)"] = {"error": repr(e)}
        try:
            out["e2e host pointers, fused consumers (H2D + kernel; never `value`)"] = host_pointer_consumers(ctx, cap, dev)
        except Exception as e:  # noqa: BLE001
            out["e2e host pointers, fused consumers (H2D + kernel; never `value`)"] = {"error": repr(e)}
        log("[bench] e2e host-pointer legs done")
        # N1 (north star): CanonicalDNAMers{31} + fx_hash over 10 Gbase LongDNA{4} on ONE GPU: 5 GB in, 160 GB out
        out.update(north_star_one_gpu(ctx, cap, stream, dev, mem, reps))
    return out


def batch_read_legs(ctx, cap, dev, seq, tseq, buf, text, a, b, timed, L, K=31):
    """Whole kmers_batch calls (layout pass, tile descriptors, element kernel; device pool, device spans, device outputs) over
    reads cut out of the bench's 1 Gbase pool -- `seq` / `buf` the LongDNA{4} words, `tseq` / `text` the same length as ASCII."""
    import numpy as np
    import torch
    res = cap.Result()
    out = {}
    rng = np.random.default_rng(20260)

    def spans_of(lengths):
        starts = np.concatenate([[0], np.cumsum(lengths[:-1])]).astype(np.int64)
        return torch.from_numpy(np.stack([starts, lengths.astype(np.int64)], axis=1).copy()).to(dev), starts

    def run(name, pool_seq, spans_t, n_reads, total, n_bases, bytes_per_base, flags):
        ms = timed(lambda: ctx.lib.kmers_batch(ctx.handle, C.byref(pool_seq), spans_t.data_ptr(), n_reads, cap.BATCH_CANONICAL, K, 2, a.data_ptr(),
                                               b.data_ptr(), 0, None, total, cap.MEM_DEVICE | cap.SPANS_DEVICE | flags, C.byref(res)))
        assert res.status == 0 and res.n_out == total, (name, res.status, res.n_out, total)
        alg = 16.0 * total + bytes_per_base * n_bases
        out[name] = {"ms": round(ms, 4), "G_elements_per_s": round(total / ms / 1e6, 1), "Gbases_per_s": round(n_bases / ms / 1e6, 1),
                     "GB_per_s": round(alg / ms / 1e6, 1), "frac_of_8TBps": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4)}

    n_reads, rl = 8_000_000, 125
    fixed = np.full(n_reads, rl, dtype=np.int64)
    spans_fixed, starts_fixed = spans_of(fixed)
    total_fixed = n_reads * (rl - K + 1)
    run(f"kmers_batch: {n_reads} reads x {rl} bases from ASCII text, CanonicalDNAMers{{31}} + fx_hash per read", tseq, spans_fixed, n_reads,
        total_fixed, n_reads * rl, 1.0, 0)
    # an N in a share of the reads: the pool's words / bytes are changed in place and put back afterwards
    for share in (0.01, 0.10):
        hit = np.flatnonzero(rng.random(n_reads) < share)
        pos = starts_fixed[hit] + rng.integers(0, rl, len(hit))
        w, sh = pos // 16, (pos % 16) * 4
        masks = np.zeros(len(w), dtype=np.uint64)
        uw, inv = np.unique(w, return_inverse=True)
        um = np.zeros(len(uw), dtype=np.uint64)
        np.bitwise_or.at(um, inv, np.uint64(0xF) << sh.astype(np.uint64))
        del masks
        idx = torch.from_numpy(uw.astype(np.int64)).to(dev)
        saved = buf[idx].clone()
        buf[idx] = torch.bitwise_or(saved, torch.from_numpy(um.view(np.int64).copy()).to(dev))
        torch.cuda.synchronize()
        run(f"kmers_batch: {n_reads} reads x {rl} bases, an N in {share:.0%} of the reads, KMERS_BATCH_SKIP (LongDNA{{4}} pool)", seq, spans_fixed,
            n_reads, total_fixed, n_reads * rl, 0.5, cap.BATCH_SKIP)
        buf[idx] = saved
        tpos = torch.from_numpy(pos.astype(np.int64)).to(dev)
        tsaved = text[tpos].clone()
        text[tpos] = 78                                                           # "N"
        torch.cuda.synchronize()
        run(f"kmers_batch: {n_reads} reads x {rl} bases from ASCII text, an N in {share:.0%} of the reads, KMERS_BATCH_SKIP", tseq, spans_fixed,
            n_reads, total_fixed, n_reads * rl, 1.0, cap.BATCH_SKIP)
        text[tpos] = tsaved
        torch.cuda.synchronize()
        del idx, saved, tpos, tsaved
    # ragged lengths: uniform 50..250 until the pool is used up
    lengths = rng.integers(50, 251, int(L / 150 * 1.02))
    lengths = lengths[np.cumsum(lengths) <= L]
    spans_rag, _ = spans_of(lengths)
    total_rag = int((lengths - K + 1).sum())
    run(f"kmers_batch: {len(lengths)} reads of 50-250 bases (uniform), LongDNA{{4}} pool", seq, spans_rag, len(lengths), total_rag, int(lengths.sum()), 0.5, 0)
    run(f"kmers_batch: {len(lengths)} reads of 50-250 bases (uniform) from ASCII text", tseq, spans_rag, len(lengths), total_rag, int(lengths.sum()), 1.0, 0)
    return out


def host_pointer_path(ctx, cap, dev, L=256_000_000, K=31):
    """The path a host WITHOUT device memory of its own takes (Julia's collect(CanonicalDNAMers{31}(seq)),
    src/iterators/CanonicalKmers.jl:220-225): kmers_canonical with KMERS_MEM_HOST -- the words go up, kmers + hashes come
    down, 16.5 bytes per kmer over PCIe.  End to end, wall clock, best of three, into pageable and into pinned host arrays;
    beside it what one plain copy of the same bytes to the same kind of host memory takes on this box (kmers_memcpy_d2h =
    hipMemcpyAsync + stream wait), and the same call as ONE launch + one copy (KMERS_PARAM_HOST_CHUNKS = -1).  Never `value`."""
    import numpy as np
    import torch

    from oracle import pyoracle
    orc = pyoracle.get()
    res = cap.Result()
    nw = (L * 4 + 63) // 64
    words = orc.synth_words(GOLDEN ^ 21, 0, nw + 1, 4)
    n = L - K + 1
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    moved = nw * 8 + 16 * n
    out = {"workload": f"CanonicalDNAMers{{{K}}} + fx_hash over {L / 1e6:.0f} Mbase LongDNA{{4}}, host pointers in and out (KMERS_MEM_HOST), "
                       f"{moved / 1e9:.2f} GB over PCIe per call"}
    dbuf = ctx.alloc(8 * n)
    try:
        for kind in ("pageable", "pinned"):
            if kind == "pageable":
                ka, ha = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
            else:
                tk, th = torch.empty(n, dtype=torch.int64, pin_memory=True), torch.empty(n, dtype=torch.int64, pin_memory=True)
                ka, ha = tk.numpy().view(np.uint64), th.numpy().view(np.uint64)
            entry = {}
            for label, chunks in (("chunked", 0), ("one_launch_one_copy", -1)):
                ctx.set_param(cap.PARAM_HOST_CHUNKS, chunks)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, ka.ctypes.data_as(C.c_void_p), ha.ctypes.data_as(C.c_void_p), 0,
                                                 cap.MEM_HOST, C.byref(res))
                    best = min(best, time.perf_counter() - t0)
                    assert rc == 0, ctx.last_error()
                entry[label] = {"ms": round(best * 1e3, 2), "Gbases_per_s": round(L / best / 1e9, 3), "PCIe_GBps": round(moved / best / 1e9, 2)}
            ctx.set_param(cap.PARAM_HOST_CHUNKS, 0)
            # the first 1 Mi elements against the oracle (the whole path is compared in tests/test_gpu_parity.py)
            ek, eh, _ = orc.canonical(words, (1 << 20) + K - 1, 4, 2, K)
            entry["verified_head"] = bool(np.array_equal(ka[:1 << 20], ek[:, 0]) and np.array_equal(ha[:1 << 20], eh))
            # the box's plain device-to-host copy into the same kind of memory (8 n bytes, best of three)
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ctx.d2h(ka, dbuf)
                best = min(best, time.perf_counter() - t0)
            d2h = 8 * n / best / 1e9
            entry["plain_d2h_copy_GBps"] = round(d2h, 2)
            entry["chunked_frac_of_plain_d2h"] = round(entry["chunked"]["PCIe_GBps"] / d2h, 4)
            out[kind] = entry
            del ka, ha
    finally:
        ctx.free(dbuf)
    return out


def host_pointer_consumers(ctx, cap, dev, L=1_000_000_000):
    """The callers that WIN from a host-memory start (missing 5): the fused consumers move 0.5-1 B/base up and almost
    nothing down -- the reference's documented uses (test/benchmark.jl:9-15; MinHash over FASTA records, docs/src/minhash.md:31-41;
    composition, docs/src/composition.md:28-39).  Each: kmers_* with KMERS_MEM_HOST over `L` symbols of a 4-bit LongSequence and of
    ASCII text, from pageable and from pinned host memory, wall clock, best of three; beside it one plain hipMemcpy of the same
    bytes from the same memory (the call cannot be faster than its input's trip), and the oracle's rate for the SAME consumer on
    one host thread and on all of them (a bounded sample).  Never `value`."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle
    orc = pyoracle.Oracle(pyoracle.build(native=True))
    res = cap.Result()
    out = {"workload": f"{L / 1e9:g} Gbase in host memory -> one call -> a word / a sketch / a table back"}
    nw4 = (L * 4 + 63) // 64
    rng = np.random.default_rng(5)
    src = {}
    pinned4 = torch.empty(nw4 + 2, dtype=torch.int64, pin_memory=True)
    w4 = pinned4.numpy().view(np.uint64)
    CH = 1 << 24
    for lo in range(0, nw4 + 1, CH):  # the synthetic sequence, in pieces (the oracle's generator is the library's)
        hi = min(nw4 + 1, lo + CH)
        w4[lo:hi] = orc.synth_words(GOLDEN ^ 22, lo, hi - lo, 4)
    src["LongDNA{4}", "pinned"] = (w4, 4, pinned4)
    src["LongDNA{4}", "pageable"] = (w4.copy(), 4, None)
    pinned8 = torch.empty(L + 16, dtype=torch.uint8, pin_memory=True)
    t8 = pinned8.numpy()
    for lo in range(0, L, 1 << 27):
        hi = min(L, lo + (1 << 27))
        t8[lo:hi] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, hi - lo, dtype=np.uint8)]
    src["ASCII text", "pinned"] = (t8, 8, pinned8)
    src["ASCII text", "pageable"] = (t8.copy(), 8, None)
    dbuf = ctx.alloc(L + 64)
    sk = np.zeros(1000, np.uint64)
    counts = np.zeros(4 ** 8, np.uint32)
    val = C.c_uint64()
    # one sketch per FASTA record (docs/src/minhash.md:31-41): the same symbols as 10 000 records of 100 kbase, spans in host memory,
    # 10 000 x 1000 hashes (80 MB) and the counts back to host memory
    n_rec, rec_len = 10_000, L // 10_000
    spans = np.stack([np.arange(n_rec, dtype=np.uint64) * np.uint64(rec_len), np.full(n_rec, rec_len, np.uint64)], axis=1).copy()
    sk_b = np.zeros((n_rec, 1000), np.uint64)
    cnt_b = np.zeros(n_rec, np.uint64)
    BATCH = f"minhash_batch(fx_hash, CanonicalDNAMers{{16}}, 1000) over {n_rec} records of {rec_len} bases"
    consumers = {
        "reduce_xor CanonicalDNAMers{31}": lambda seq: ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_HOST, C.byref(res)),
        "minhash(fx_hash, CanonicalDNAMers{16}, 1000)": lambda seq: ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p),
                                                                                          cap.MEM_HOST, C.byref(res)),
        "composition FwDNAMers{8}": lambda seq: ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 8, counts.ctypes.data_as(C.c_void_p), cap.MEM_HOST, C.byref(res)),
        BATCH: lambda seq: ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans.ctypes.data_as(C.c_void_p), n_rec, 16, 2, 0, 1000,
                                                       sk_b.ctypes.data_as(C.c_void_p), cnt_b.ctypes.data_as(C.c_void_p), cap.MEM_HOST, C.byref(res)),
    }
    try:
        for (what, kind), (arr, bits, _keep) in src.items():
            nbytes = (L * bits + 7) // 8
            seq = cap.Seq(arr.ctypes.data, L, 0, 0, bits, 0)
            best = 1e9
            for _ in range(3):  # the plain copy of the same bytes from the same memory
                t0 = time.perf_counter()
                ctx.check(ctx.lib.kmers_memcpy_h2d(ctx.handle, C.c_void_p(dbuf), arr.ctypes.data_as(C.c_void_p), nbytes), "kmers_memcpy_h2d")
                best = min(best, time.perf_counter() - t0)
            h2d = nbytes / best / 1e9
            entry = {"bytes_up_GB": round(nbytes / 1e9, 3), "plain_h2d_copy_GBps": round(h2d, 2)}
            for name, call in consumers.items():
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    rc = call(seq)
                    best = min(best, time.perf_counter() - t0)
                    assert rc == 0, ctx.last_error()
                entry[name] = {"ms": round(best * 1e3, 2), "Gbases_per_s": round(L / best / 1e9, 2), "frac_of_plain_h2d": round(nbytes / best / 1e9 / h2d, 4)}
            out[f"{what}, {kind} host memory"] = entry
        # the same consumers on the host cores (the oracle: C restatement of the reference, -O3 -march=native), 16 Mbase per thread
        S = 1 << 24
        ws = orc.synth_words(GOLDEN ^ 22, 0, S // 16 + 1, 4)
        ncpu = len(os.sched_getaffinity(0))

        def xor_cpu(_i=0):
            return orc.reduce_xor_canonical(ws, S, 4, 2, 31)[0]

        def minhash_cpu(_i=0):  # sketch(fx_hash, CanonicalDNAMers{16}(seq), 1000), docs/src/minhash.md:31-35: hashes, then the 1000 smallest distinct
            _, eh, _ = orc.canonical(ws, S, 4, 2, 16)
            return np.unique(np.partition(eh, 4000)[:4000])[:1000]

        def comp_cpu(_i=0):     # counts[as_integer(kmer) + 1] += 1 over FwDNAMers{8}, docs/src/composition.md:28-39
            fw, _ = orc.fw_kmers(ws, S, 4, 2, 8)
            return np.bincount(fw[:, 0].astype(np.int64), minlength=4 ** 8)

        def batch_cpu(_i=0):    # the per-record loop of docs/src/minhash.md:31-41 over the sample cut into records of the leg's length
            outs = []               # (4-bit symbols: record r begins on a word boundary, rec_len * 4 / 64 words in)
            assert rec_len * 4 % 64 == 0
            for r in range(S // rec_len):
                _, eh, _ = orc.canonical(ws[r * rec_len // 16:], rec_len, 4, 2, 16)
                outs.append(np.unique(np.partition(eh, 4000)[:4000])[:1000])
            return outs
        cpu = {}
        for name, fn in (("reduce_xor CanonicalDNAMers{31}", xor_cpu), ("minhash(fx_hash, CanonicalDNAMers{16}, 1000)", minhash_cpu),
                         ("composition FwDNAMers{8}", comp_cpu), (BATCH, batch_cpu)):
            t0 = time.perf_counter()
            fn()
            one = S / (time.perf_counter() - t0) / 1e9
            t0 = time.perf_counter()
            with ThreadPoolExecutor(ncpu) as ex:
                list(ex.map(fn, range(ncpu)))
            allc = ncpu * S / (time.perf_counter() - t0) / 1e9
            cpu[name] = {"one_thread_Gbases_per_s": round(one, 3), "all_cores_Gbases_per_s": round(allc, 3), "cores": ncpu}
        # the GPU's answers against the host's on the sample: XOR word, sketch, table
        seq_s = cap.Seq(ws.ctypes.data, S, 0, 0, 4, 0)
        ok = consumers["reduce_xor CanonicalDNAMers{31}"](seq_s) == 0 and val.value == xor_cpu()
        ok &= consumers["minhash(fx_hash, CanonicalDNAMers{16}, 1000)"](seq_s) == 0 and bool(np.array_equal(sk, minhash_cpu()))
        ok &= consumers["composition FwDNAMers{8}"](seq_s) == 0 and bool(np.array_equal(counts, comp_cpu().astype(np.uint32)))
        n_s = S // rec_len  # (the batch call over the sample's records: the first n_s spans are the sample's)
        rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_s), spans.ctypes.data_as(C.c_void_p), n_s, 16, 2, 0, 1000, sk_b.ctypes.data_as(C.c_void_p),
                                         cnt_b.ctypes.data_as(C.c_void_p), cap.MEM_HOST, C.byref(res))
        ok &= rc == 0 and all(int(cnt_b[r]) == len(e) and bool(np.array_equal(sk_b[r, :len(e)], e)) for r, e in enumerate(batch_cpu()))
        out["cpu_port_same_consumers"] = dict(cpu, sample=f"{S >> 20} Mi-base of the same generator per thread, oracle (C restatement, gcc -O3 -march=native) + numpy for "
                                                           "the sketch's selection and the table", verified_against_gpu=bool(ok))
    finally:
        ctx.free(dbuf)
    return out


def north_star_one_gpu(ctx, cap, stream, dev, mem, reps=7, L=NORTH_STAR_BASES):
    import torch
    res = cap.Result()
    K = 31
    n = L - K + 1
    need = 8 * (2 * n + L // 16 + 2)
    room = mem.room()
    name = f"N1 north star: CanonicalDNAMers{{31}} + fx_hash, {L / 1e9:g} Gbase LongDNA{{4}}, one GPU, 16.5 B/kmer"
    if room < need + (6 << 30):
        return {name: {"skipped": f"needs {need / 1e9:.0f} GB of HBM, {room / 1e9:.0f} GB available"}}
    seed10 = GOLDEN ^ 10
    nw = (L * 4 + 63) // 64
    a, h = mem.empty(n), mem.empty(n)  # (the two 80 GB arrays one after the other: different classes at every position)
    buf = mem.empty(nw + 2)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed10, 0, nw, 4, 0, buf.data_ptr()), "kmers_synth_dna")
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    ms = busy_timed(ctx, stream, lambda: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, a.data_ptr(), h.data_ptr(), 0, ASYNC, C.byref(res)), reps)
    ok = verify_canonical(ctx, cap, stream, dev, buf, L, 0, 0, 4, K, 1, seed10, a, h, n)
    mem.free(buf, a, h)
    alg = 16.5 * n
    return {name: {"kernel_ms": round(ms, 4), "Gbases_per_s": round(L / ms / 1e6, 1), "GB_per_s": round(alg / ms / 1e6, 1),
                   "frac_of_8TBps": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4), "verified": ok}}


# --------------------------------------------------------------------------------------------
# PMC: HBM bytes of the headline kernel (roofline.traffic) and VALU issue shares of the legs that are not pure store
# streams, measured in THIS run by profiled child processes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC slots,
# MI355X_MICROARCH.md "rocprofv3 PMC slots")
def pmc_child(args):
    """The profiled program (run under rocprofv3 --pmc): "headline" = the headline launch three times, nothing else;
    "legs" = UnambiguousKmers on the C5 lattice and at K = 31, then the fused consumers, twice each, in this order."""
    import numpy as np
    import torch

    import kmers_jl_amd as km
    cap = km._capi
    ctx = km.Context(0)
    dev = torch.device("cuda", 0)
    stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
    K, bits, L = args.k, args.src_bits, args.bases
    res = cap.Result()

    def synth(seed, amb=0):
        nw = (L * bits + 63) // 64
        with torch.cuda.stream(stream):
            buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
            ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, 0, nw, bits, amb, buf.data_ptr()), "kmers_synth_dna")
        return buf
    if args.pmc_child == "headline":
        # the outputs from this process's own pool, as in the timed leg, so that the launcher sees placed arrays and picks the same
        # shape; the shape it picked goes to stdout for the parent
        n = L - K + 1
        N = cap.load().kmers_words_per_kmer(K, 2)
        if args.alloc == "pool":
            p_k = ctx.alloc(8 * n * N, lone_output=args.no_hash)
            p_h = None if args.no_hash else ctx.alloc(8 * n)
        else:
            out_k = torch.empty(n * N, dtype=torch.int64, device=dev)
            out_h = None if args.no_hash else torch.empty(n, dtype=torch.int64, device=dev)
            p_k, p_h = out_k.data_ptr(), (out_h.data_ptr() if out_h is not None else None)
        buf = synth(GOLDEN ^ 2)
        seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
        for _ in range(3):
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, p_k, p_h, 0, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0, ctx.last_error()
        torch.cuda.synchronize()
        t, tile, split = ctx.last_launch_shape()
        print(f"PMC_CHILD_SHAPE {t}x{tile}" + (" split" if split else ""), flush=True)
        return
    a = torch.empty(L, dtype=torch.int64, device=dev)
    b = torch.empty(L, dtype=torch.int64, device=dev)
    amb = synth(GOLDEN ^ 5, 2621)
    seqa = cap.Seq(amb.data_ptr(), L, 0, 0, bits, 0)
    for Ku, Ju in ((21, 3), (31, 1)):
        for _ in range(2):
            rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqa), Ku, Ju, a.data_ptr(), b.data_ptr(), L, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0, ctx.last_error()
    del amb
    clean = synth(GOLDEN ^ 5)
    seq = cap.Seq(clean.data_ptr(), L, 0, 0, bits, 0)
    val = C.c_uint64()
    sk = np.zeros(1000, dtype=np.uint64)
    for _ in range(2):
        assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), 31, 2, 1, C.byref(val), cap.MEM_DEVICE, C.byref(res)) == 0
    for _ in range(2):
        assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 16, 2, 0, 1000, sk.ctypes.data_as(C.c_void_p), cap.MEM_DEVICE, C.byref(res)) == 0
    for Kc in (4, 8):
        for _ in range(2):
            assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), Kc, b.data_ptr(), cap.MEM_DEVICE, C.byref(res)) == 0
    torch.cuda.synchronize()


def profiler_in_environment():
    """True when this process itself runs under a profiler (rocprofv3 preloads its tool library through these variables).
    A nested rocprofv3 would inherit them, and its launcher -- with the GPU already initialised by the preloaded tool --
    would exec the child program: on this pool a process that has touched the GPU must never replace itself."""
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER_")) for k in os.environ):
        return True
    return any(s in os.environ.get("LD_PRELOAD", "") for s in ("rocprof", "roctracer", "rocprofiler"))


def run_pmc_pass(args, which, counters, device_index, timeout):
    """One `rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --pmc-child <which>` child.  Returns
    ([(kernel name, dispatch id, counter, value)], [(kernel name, dispatch id, duration in ns)]) or raises."""
    import csv
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    tmp = tempfile.mkdtemp(prefix="kmers_pmc_")
    try:
        cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", tmp, "--", sys.executable,
               os.path.abspath(__file__), "--pmc-child", which, "--bases", str(args.bases), "--k", str(args.k), "--src-bits", str(args.src_bits)]
        if args.no_hash:
            cmd.append("--no-hash")
        cmd += ["--alloc", getattr(args, "alloc", "pool")]
        # a clean environment for the child: no profiler variables of an outer run, one visible device (the rank's own)
        env = {k: v for k, v in os.environ.items()
               if not k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER_")) and k not in ("LD_PRELOAD", "RANK", "LOCAL_RANK", "WORLD_SIZE",
                                                                                        "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        visible = os.environ.get("HIP_VISIBLE_DEVICES")
        env["HIP_VISIBLE_DEVICES"] = visible.split(",")[device_index] if visible else str(device_index)
        env["TMPDIR"] = tmp
        p = subprocess.run(cmd, cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        if p.returncode != 0:
            raise RuntimeError(f"rocprofv3 --pmc {' '.join(counters)} child failed ({p.returncode}): {p.stderr[-300:]}")
        rows, durs = [], []
        for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                rows.append((r.get("Kernel_Name", ""), int(r.get("Dispatch_Id", 0)), r.get("Counter_Name"), float(r["Counter_Value"])))
        for f in glob.glob(os.path.join(tmp, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                durs.append((r.get("Kernel_Name", ""), int(r.get("Dispatch_Id", 0)), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
        shape = [l.split(" ", 1)[1] for l in p.stdout.splitlines() if l.startswith("PMC_CHILD_SHAPE ")]
        return rows, durs, (shape[-1] if shape else None)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_traffic(args, device_index=0, timeout=600):
    """Returns (bytes per launch or None, description of where the number comes from)."""
    if profiler_in_environment():
        return None, "this run is itself under a profiler (LD_PRELOAD / ROCP_* set): no nested rocprofv3"
    vals = {}
    shape = None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            rows, _, shape = run_pmc_pass(args, "headline", [counter], device_index, timeout)
            v = [x for (name, _, cn, x) in rows if "stream_kernel" in name and cn == counter]
            if not v:
                return None, f"no {counter} rows for stream_kernel in the rocprofv3 output"
            vals[counter] = sum(v) / len(v)
    except Exception as e:
        return None, f"PMC pass failed: {e!r}"
    # units and gfx950 corrections exactly as MI355X_MICROARCH.md (HBM section) prescribes: both counters are in KiB;
    # FETCH_SIZE reports half of a coalesced streaming read on gfx950 -> doubled; WRITE_SIZE is exact for 16 B/lane stores
    traffic = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return traffic, (f"measured in this run: two rocprofv3 --pmc child passes, each a fresh process that launches the headline kernel at the "
                     f"headline size into outputs from a pool of its own (launch shape there: {shape}) (FETCH_SIZE {vals['FETCH_SIZE']:.1f} KiB "
                     f"x 2 [gfx950 correction], WRITE_SIZE {vals['WRITE_SIZE']:.1f} KiB)")


def measure_legs(args, device_index=0, timeout=600):
    """Vector-instruction account of the legs whose roofline is not (only) HBM, from one rocprofv3 --pmc child over the `legs`
    program: per leg ONE dispatch's duration (kernel trace), SQ_INSTS_VALU and GRBM_GUI_ACTIVE (/ 8 XCDs = active cycles, so
    clock = cycles / duration).  Returns {leg: {...}} (empty when the pass could not run).  The legs are told apart by kernel
    name and dispatch order."""
    if profiler_in_environment():
        return {}
    try:
        rows, durs, _ = run_pmc_pass(args, "legs", ["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"], device_index, timeout)
    except Exception as e:
        log(f"VALU pass failed: {e!r}")
        return {}
    per = {}
    for name, did, cn, x in rows:
        per.setdefault(did, {"name": name})[cn] = per.get(did, {}).get(cn, 0.0) + x
    for name, did, ns in durs:
        if did in per:
            per[did]["ns"] = ns
    order = [per[d] for d in sorted(per)]

    def pick(sub, excl=()):
        return [d for d in order if sub in d["name"] and not any(e in d["name"] for e in excl) and "SQ_INSTS_VALU" in d]
    # kmers_unambiguous launches its one-pass kernel once per call here (capacity given): u21 twice, then u31 twice
    un = pick("unambiguous_kernel")
    groups = {"u21": un[0:2], "u31": un[2:4]}
    rk = pick("run_kernel")
    groups["xor"], groups["minhash"] = rk[0:2], rk[2:4]
    comp = pick("composition_kernel")
    groups["comp4"], groups["comp8"] = comp[0:2], comp[2:4]
    res = {}
    for leg, ds in groups.items():
        if not ds:
            continue
        d = ds[-1]  # ONE dispatch: its duration, its instruction count and its active cycles
        valu, act = d["SQ_INSTS_VALU"], d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        ns = d.get("ns", 0)
        if not (valu and act and ns):
            continue
        clock_ghz = act / ns
        res[leg] = {"valu_insts": int(valu), "profiled_ms": round(ns / 1e6, 4), "clock_GHz": round(clock_ghz, 3),
                    "cycles_per_inst": round(act * N_SIMDS / valu, 3),
                    "valu_floor_ms": round(valu * VALU_CYCLES_FAST / N_SIMDS / (clock_ghz * 1e6), 4),
                    "valu_floor_ms_slow": round(valu * VALU_CYCLES_SLOW / N_SIMDS / (clock_ghz * 1e6), 4),
                    "source": f"one dispatch of a rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace child pass of this run, kernel {d['name'][:60]}; "
                              f"issue rates {VALU_CYCLES_FAST} / {VALU_CYCLES_SLOW} cycles per instruction and SIMD: tools/device_probes/valu_rates.hip, profiles/r04_valu_rates.txt"}
    return res


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
class Leg:
    """One sharded canonical(+hash) workload resident on this rank: its shard of `total_bases`, the halo step and the launch."""

    def __init__(self, env, total_bases, seed):
        import torch
        self.env = env
        ctx, cap, mem, args = env.ctx, env.cap, env.mem, env.args
        from kmers_jl_amd.shard import HaloExchanger, plan_shards
        self.total_bases, self.seed = total_bases, seed
        self.plan = plan_shards(total_bases, args.k, env.world, args.src_bits)
        sh = self.sh = self.plan[env.rank]
        self.N = cap.load().kmers_words_per_kmer(args.k, 2)
        with torch.cuda.stream(env.stream):
            # the two output arrays first and one right after the other: the library gives consecutive blocks different region classes
            self.out_k = mem.empty(sh.n_kmers * self.N, lone_output=args.no_hash)
            self.out_h = None if args.no_hash else mem.empty(sh.n_kmers)
            self.buf = mem.empty(sh.n_own_words + sh.halo_words + 2)
            self.buf.zero_()
            ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words, args.src_bits, 0, self.buf.data_ptr()),
                      "kmers_synth_dna")
            self.halo = None
            if env.grouped and env.transport != "native":
                self.halo = HaloExchanger(self.buf, sh, self.plan, transport=env.transport)  # its workspace is filled on this stream too
        self.shard_c = env.comm._shard_struct(sh) if env.comm is not None else None
        torch.cuda.synchronize()
        self.seq = cap.Seq(self.buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, args.src_bits, 0)
        self.res = cap.Result()
        self.ph = self.out_h.data_ptr() if self.out_h is not None else None

    def step(self, ev=None):
        import torch
        env = self.env
        ctx, cap = env.ctx, env.cap
        exchanges = env.comm is not None or self.halo is not None
        with torch.cuda.stream(env.stream):
            if ev and exchanges:   # (an event record is a marker packet of its own on the stream: none where there is nothing to time)
                ev[0].record(env.stream)
            if env.comm is not None:
                env.comm.halo_exchange(self.shard_c, self.buf.data_ptr())
            elif self.halo is not None:
                self.halo.exchange()
            if ev and ev[1] is not None:
                ev[1].record(env.stream)
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(self.seq), env.args.k, 2, self.out_k.data_ptr(), self.ph, 0,
                                         cap.MEM_DEVICE | cap.ASYNC, C.byref(self.res))
            if ev:
                ev[2].record(env.stream)
        if rc != 0:
            raise RuntimeError(f"kmers_canonical failed: {ctx.last_error()}")

    def timed(self, warmup, steps, solo=False):
        """W untimed steps, then exactly `steps` steps between two fences (device sync + barrier + device sync); returns
        (elapsed seconds, mean kernel ms, mean halo-step ms) of THIS rank.  solo: this rank runs alone (no barrier)."""
        import numpy as np
        import torch
        env = self.env
        for _ in range(warmup):
            self.step()
        rc, _ = env.ctx.sync()
        assert rc == 0, env.ctx.last_error()
        # Every event record is a marker packet of its own on the stream (5 us of a 2.36 ms step each, measured: three per step cost the
        # timed region 0.5 %).  With a halo step in front of the launch: three per step (halo | kernel).  Without one: ONE per step -- the
        # mark behind launch i is the mark in front of launch i + 1, a launch's duration is the interval between its two marks (the
        # few microseconds between two launches included: what the stream spends per launch).
        exchanges = env.comm is not None or self.halo is not None
        if exchanges:
            events = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps)]
        else:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            events = [(None, marks[0] if i == 0 else None, marks[i + 1]) for i in range(steps)]
        env.fence(solo)
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(events[i])
        env.fence(solo)
        elapsed = time.perf_counter() - t0
        rc, _ = env.ctx.sync()
        assert rc == 0, env.ctx.last_error()
        if exchanges:
            kern = float(np.mean([e[1].elapsed_time(e[2]) for e in events])) if steps else 0.0
            halo = float(np.mean([e[0].elapsed_time(e[1]) for e in events])) if steps else 0.0
        else:
            kern = float(np.mean([marks[i].elapsed_time(marks[i + 1]) for i in range(steps)])) if steps else 0.0
            halo = 0.0
        return elapsed, kern, halo

    def verify(self):
        env, sh, args = self.env, self.sh, self.env.args
        return verify_canonical(env.ctx, env.cap, env.stream, env.dev, self.buf, sh.n_bases, sh.first_kmer, sh.first_word, args.src_bits,
                                args.k, self.N, self.seed, self.out_k, self.out_h, sh.n_kmers)

    def release(self):
        import torch
        self.env.mem.free(self.buf, self.out_k, self.out_h)
        self.buf = self.out_k = self.out_h = self.halo = None
        torch.cuda.empty_cache()


def fresh_outputs_leg(env, leg, warmup, steps, resident_ms_per_step):
    """The timed region again with FRESH outputs for every step: {kmers_dev_alloc(kmers), kmers_dev_alloc(hashes), launch,
    kmers_dev_free, kmers_dev_free} x steps between two fences, same source, same wall clock -- what a host that calls
    `collect(CanonicalDNAMers{K}(seq))` per sequence pays for its allocator (the headline's allocator must be usable
    that way).  Host time inside the calls, and what the pool did, are reported beside it."""
    import numpy as np
    ctx, cap, args = env.ctx, env.cap, env.args
    n, N = leg.sh.n_kmers, leg.N
    res = cap.Result()
    flags = cap.MEM_DEVICE | cap.ASYNC
    alloc_us, free_us = [], []

    def one(timed):
        t0 = time.perf_counter()
        pk = ctx.alloc(8 * n * N, lone_output=args.no_hash)
        ph = None if args.no_hash else ctx.alloc(8 * n)
        t1 = time.perf_counter()
        rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(leg.seq), args.k, 2, pk, ph, 0, flags, C.byref(res))
        t2 = time.perf_counter()
        ctx.free(pk)
        if ph:
            ctx.free(ph)
        t3 = time.perf_counter()
        if rc != 0:
            raise RuntimeError(f"kmers_canonical failed: {ctx.last_error()}")
        if timed:
            alloc_us.append((t1 - t0) * 1e6)
            free_us.append((t3 - t2) * 1e6)
    t_first = time.perf_counter()
    one(False)  # the first round assembles the blocks (the pool walks for its classes, maps, checks): priced on its own
    env.fence(True)
    first_ms = (time.perf_counter() - t_first) * 1e3
    for _ in range(warmup):
        one(False)
    before = ctx.pool_stats()
    env.fence(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        one(True)
    env.fence(True)
    elapsed = time.perf_counter() - t0
    rc, _ = ctx.sync()
    assert rc == 0, ctx.last_error()
    after = ctx.pool_stats()
    ms = elapsed / steps * 1e3
    return {"what": "the K timed steps again with fresh outputs per step: {kmers_dev_alloc x 2, launch, kmers_dev_free x 2} between two fences "
                    "(the resident arrays of the headline stay allocated beside them)",
            "ms_per_step": round(ms, 4), "Gbases_per_s": round(leg.sh.n_bases / ms / 1e6, 2),
            "over_resident": round(ms / resident_ms_per_step, 4),
            "alloc_us_per_step": round(float(np.median(alloc_us)), 1), "free_us_per_step": round(float(np.median(free_us)), 1),
            "first_round_ms": round(first_ms, 2),
            "pool_cache_hits": after["cache_hits"] - before["cache_hits"], "pool_blocks_assembled": after["cache_misses"] - before["cache_misses"],
            "pool_handles_created": after["chunks_created"] - before["chunks_created"]}


class Env:
    pass


def gather_floats(env, values):
    """[values of rank 0, values of rank 1, ...] (each a list of floats) on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(values, dtype=torch.float64, device=env.dev)
    if not env.grouped:
        return [list(map(float, t.tolist()))]
    parts = [torch.zeros_like(t) for _ in range(env.world)]
    dist.all_gather(parts, t)
    return [list(map(float, p.tolist())) for p in parts]


def strong_scaling_entry(args, K, bits, world, strong_bases, s_per_rank, bytes_per_kmer, verified, solo):
    """The `strong_scaling` object of a weak N > 1 line: the north-star input split over the N ranks (per-rank [elapsed s, kernel
    ms, halo-step ms, kmers]) and, when rank 0 also ran it alone (`solo` = [elapsed s, kernel ms]), the ratio of the two."""
    tN = max(p[0] for p in s_per_rank) / args.steps
    entry = {
        "workload": f"CanonicalDNAMers{{{K}}} + fx_hash over ONE sequence of {strong_bases / 1e9:g} Gbase LongDNA{{{bits}}} split over {world} GPUs "
                    f"(kmers_shard_plan: contiguous kmer ranges, (K-1)-base halo), same steps / warm-up / fences as the headline",
        "scaling": "strong", "total_bases": strong_bases, "n_gpus": world,
        "ms_per_step": round(tN * 1e3, 4), "value": round(strong_bases / tN / 1e9, 3), "unit": "Gbases/s",
        "kernel_ms_per_rank": [round(p[1], 4) for p in s_per_rank], "halo_ms_per_rank": [round(p[2], 4) for p in s_per_rank],
        "frac_per_rank": [round(bytes_per_kmer * p[3] / p[1] / 1e6 / HBM_PEAK_GBPS, 4) if p[1] else None for p in s_per_rank],
        "verified": verified,
    }
    if solo and solo[0]:
        t1 = solo[0] / args.steps
        entry["one_gpu"] = {"ms_per_step": round(t1 * 1e3, 4), "value": round(strong_bases / t1 / 1e9, 3), "kernel_ms": round(solo[1], 4),
                            "what": "the same input on rank 0's GPU alone, same run, same steps"}
        entry["speedup_vs_one_gpu"] = round(t1 / tN, 3)
    return entry


def assemble_line(args, K, bits, world, strong, grouped, backend, transport, total_bases, plan_bases, seed, elapsed, per_rank, bytes_per_kmer,
                  verified, use_lib, write_ceiling, shape_report, plain_alloc, fill_gbps, strong_extra):
    """Rank 0's JSON line from what the ranks measured (per_rank[r] = [elapsed s, kernel ms, halo-step ms, kmers]); pure, so that
    tests/test_bench_contract.py can check the N > 1 schema on a machine without GPUs.  roofline.traffic, other_configs and
    cpu_baseline are filled in by the caller."""
    kern_list = [p[1] for p in per_rank]
    halo_list = [p[2] for p in per_rank]
    fracs = [bytes_per_kmer * p[3] / p[1] / 1e6 / HBM_PEAK_GBPS if p[1] else 0.0 for p in per_rank]
    worst = min(range(len(fracs)), key=fracs.__getitem__)  # the slowest GPU bounds the job
    achieved = fracs[worst] * HBM_PEAK_GBPS
    if not grouped:
        sharding = "single shard"
    else:
        how = {"native": "kmers_halo_exchange of the C ABI: grouped ncclSend/ncclRecv on the kernel's stream (RCCL; torch carried only the ncclUniqueId)",
               "allgather": f"torch.distributed all_gather of <= 32 B per rank ({backend})",
               "p2p": f"torch.distributed batch_isend_irecv between neighbours ({backend})"}[transport]
        sharding = f"contiguous kmer-start ranges, (K-1)-base halo from rank+1 each step; backend {backend}; transport {transport}: {how}"
    what = f"CanonicalDNAMers{{{K}}}" + ("" if args.no_hash else " + fx_hash")
    if strong:
        workload = (f"{what} over ONE sequence of {total_bases / 1e9:g} Gbase LongDNA{{{bits}}} split over {world} GPU(s) "
                    f"(north star: 10 Gbase LongDNA{{4}}), kmers and hashes materialised in HBM")
    else:
        workload = (f"{what} over {args.bases / 1e9:g} Gbase LongDNA{{{bits}}} per GPU (BASELINE.json configs[1]), "
                    f"kmers and hashes materialised in HBM")
    line = {
        "metric": METRIC, "value": round(total_bases * args.steps / elapsed / 1e9, 3), "unit": "Gbases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "k": K, "src_bits": bits, "total_bases": total_bases,
                   "bases_per_gpu": plan_bases if strong else args.bases,
                   "sharding": sharding, "backend": backend, "halo_transport": transport if grouped else None,
                   "seed": hex(seed), "wake_s": args.wake_s,
                   "alloc": ("kmers_dev_alloc: the device's class pool (no reservation)" if use_lib else "torch.empty (hipMalloc)")},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "traffic_source": "not measured",
                     "kernel": "stream_kernel<src_bits,N,CANON,stride1>", "kernel_ms": round(kern_list[worst], 4),
                     "bytes_per_kmer": bytes_per_kmer, "kmers_per_launch": int(per_rank[worst][3])},
        "verified": verified,
    }
    rf = line["roofline"]
    if grouped:
        rf["per_gpu"] = "achieved / frac / kernel_ms are those of the slowest GPU; the lists are in rank order"
        rf["kernel_ms_per_rank"] = [round(x, 4) for x in kern_list]
        rf["kernel_ms_min"], rf["kernel_ms_max"] = round(min(kern_list), 4), round(max(kern_list), 4)
        rf["frac_per_rank"] = [round(x, 4) for x in fracs]
        rf["halo_step_ms_per_rank"] = [round(x, 4) for x in halo_list]
        rf["halo_step_share"] = round(max(halo_list) / max(1e-9, max(halo_list) + max(kern_list)), 5)
    rf["write_ceiling_GBps"] = round(write_ceiling[0], 1)
    rf["write_ceiling_source"] = write_ceiling[1]
    rf["frac_of_write_ceiling"] = round(achieved / write_ceiling[0], 4)
    if shape_report is not None:
        rf["launch_shape"] = shape_report
    if plain_alloc is not None:
        rf["plain_alloc"] = plain_alloc
    if fill_gbps:
        rf["torch_fill_GBps"] = round(fill_gbps, 1)  # torch.Tensor.fill_ over the same two arrays, same run
        rf["vs_torch_fill"] = round(achieved / fill_gbps, 4)
    if strong_extra is not None:
        line["strong_scaling"] = strong_extra
    return line


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
        # rank 0 works alone for a while after the timed region (PMC child passes, the CPU baseline, the 1-GPU leg of the
        # strong-scaling pair) while the others wait in a barrier: give the collectives' watchdog room for that
        long_wait = datetime.timedelta(minutes=45)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=long_wait)
        else:
            dist.init_process_group(backend, timeout=long_wait)
        backend = dist.get_backend()
    world = dist.get_world_size() if grouped else 1
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: the process group has {world} ranks")

    import kmers_jl_amd as km
    from kmers_jl_amd.shard import NativeComm
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
    strong = args.total_bases > 0
    total_bases = args.total_bases if strong else args.bases * world
    seed = GOLDEN ^ 2  # golden ^ config id (C2)
    N = cap.load().kmers_words_per_kmer(K, 2)
    bytes_per_kmer = bits / 8 + 8 * N + (0 if args.no_hash else 8)

    # the halo transport: under RCCL the C ABI's own exchange (grouped ncclSend/ncclRecv on the context's stream; torch only
    # hands the 128-byte ncclUniqueId around); under gloo (shared-device debugging, CPU tests) torch.distributed's
    transport = os.environ.get("KMERS_HALO_TRANSPORT", "native" if backend == "nccl" else "allgather")
    shared_device = grouped and backend != "nccl"  # (gloo debugging mode: the ranks share one device and its memory)
    use_pool = args.alloc == "pool" and not shared_device  # (the library's allocator: the device's class pool)

    env = Env()
    env.args, env.ctx, env.cap, env.dev, env.stream = args, ctx, cap, dev, stream
    env.rank, env.world, env.grouped, env.transport, env.comm = rank, world, grouped, transport, None

    def fence(solo=False):
        torch.cuda.synchronize()
        if grouped and not solo:
            dist.barrier()
        torch.cuda.synchronize()
    env.fence = fence

    # ---- the same launch into PLAIN allocations, before the pool exists (rank 0, N = 1; not the headline) -----------
    # Where the outputs live is worth 8-10 % on this device (profiles/r03_alloc.md, r05_vmm.md): this is what a host gets that
    # allocates its outputs one hipMalloc each on a fresh machine, kept in the line next to the pool's figure.
    plain_alloc = None
    if use_pool and world == 1 and not strong and not args.no_other_configs:
        try:
            env.mem = Memory(ctx, dev, False)
            leg0 = Leg(env, total_bases, seed)
            ms0 = busy_timed(ctx, stream, leg0.step, reps=7, busy_s=0.3)
            plain_alloc = {"kernel_ms": round(ms0, 4), "frac": round(bytes_per_kmer * leg0.sh.n_kmers / ms0 / 1e6 / HBM_PEAK_GBPS, 4),
                           "what": "the headline launch into two torch (hipMalloc) allocations made before the library's allocator was used; "
                                   "7 launches behind 0.3 s of the same launch, outside the timed region"}
            del leg0
            torch.cuda.empty_cache()
        except Exception as e:
            plain_alloc = {"error": repr(e)}
    write_ceiling = (HBM_PEAK_GBPS, "8 TB/s spec (no pool in this run)")
    mem = env.mem = Memory(ctx, dev, use_pool)
    if grouped and transport == "native":
        if backend != "nccl":
            raise SystemExit("KMERS_HALO_TRANSPORT=native needs one GPU per rank (RCCL); the gloo mode shares a device")
        # If the library's own communicator cannot be built on this machine (no RCCL the library can load, an
        # ncclCommInitRank that fails) the run goes on with torch.distributed's RCCL for the halo -- said in the line
        # (halo_transport) and on stderr -- rather than losing the measurement.  All ranks take the same decision.
        try:
            env.comm = NativeComm.bootstrap(ctx)
            made = 1
        except Exception as e:  # noqa: BLE001
            sys.stderr.write(f"[bench rank {rank}] kmers_comm_create failed ({e!r})\n")
            made = 0
        agreed = torch.tensor([made], dtype=torch.int32, device=dev)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            if env.comm is not None:
                env.comm.close()
                env.comm = None
            transport = env.transport = "allgather"
            if rank == 0:
                sys.stderr.write("[bench] halo transport falls back to torch.distributed all_gather over RCCL\n")
    comm = env.comm

    leg = Leg(env, total_bases, seed)
    sh, plan = leg.sh, leg.plan
    pool_report = None
    if use_pool:  # what the pool made of the two arrays, and the write ceiling ITS probes measured on this box
        info = ctx.pool_info()
        runs = lambda p: " ".join(f"{'ABCD?'[c]}{n}" for c, n in run_lengths(ctx.pool_layout(p)[1])) if p else None
        pool_report = {"held_GB": round(info["held"] / 1e9, 1), "in_use_GB": round(info["in_use"] / 1e9, 1),
                       "held_over_in_use": round(info["held"] / max(1, info["in_use"]), 3), "classes": info["n_classes"],
                       "GB_per_class": [round(b / 1e9, 1) for b in info["class_bytes"]],
                       "kmers_array": runs(leg.out_k.data_ptr()), "hashes_array": runs(leg.out_h.data_ptr() if leg.out_h is not None else 0),
                       "what": "1 GiB handles of HIP virtual-memory management, class of each measured by the pool; arrays as runs of handles per class"}
        if info["two_class_gbps"] > 0:
            write_ceiling = (info["two_class_gbps"], f"kmers_pool_info: the fastest probe of this run's pool, two 1 GiB store streams side by side in two "
                                                     f"region classes ({info['two_class_gbps']:.0f} GB/s; inside one class {info['one_class_gbps']:.0f} GB/s)")

    # Wake the device: after an idle gap (allocation, data generation, process start) this device runs its next ~10 ms
    # 10-25 % slower (profiles/r02_tuning.md section 1) and the first second of a fresh process 2 % slower than the ninety
    # that follow, which is longer than the W warm-up steps of the contract.  --wake-s seconds (default 1) of plain torch
    # fills of the output arrays -- not steps of the hot path -- come first; the W warm-up steps and the K timed steps
    # are exactly the contract's.
    with torch.cuda.stream(stream):
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < args.wake_s:
            leg.out_k.fill_(0)
            if leg.out_h is not None:
                leg.out_h.fill_(0)
            torch.cuda.synchronize()
    log("[bench] buffers ready, device awake: the timed region")
    elapsed, kern_ms, halo_ms = leg.timed(args.warmup, args.steps)
    per_rank = gather_floats(env, [elapsed, kern_ms, halo_ms, float(sh.n_kmers)])
    elapsed = max(p[0] for p in per_rank)

    chosen_shape = ctx.last_launch_shape()
    fresh_report = None
    if use_pool and rank == 0 and world == 1 and not strong:
        try:
            fresh_report = fresh_outputs_leg(env, leg, args.warmup, args.steps, elapsed / args.steps * 1e3)
        except Exception as e:  # noqa: BLE001
            fresh_report = {"error": repr(e)}
    # ---- integrity of what the timed kernel wrote (outside the timed region) --------------
    log("[bench] timed region and fresh-outputs leg done")
    verified = leg.verify()
    # ---- the launcher's choice against the alternatives of its table (stream_launch.hpp), same arrays, same run --------
    shape_report = None
    if rank == 0 and world == 1 and not args.no_other_configs and not args.tile:
        try:
            cands = {}
            for t, tile in ((128, 1536), (256, 1024), (128, 1024), (256, 1536), (256, 2048)):
                ctx.set_param(cap.PARAM_BLOCK_THREADS, t)
                ctx.set_param(cap.PARAM_TILE_KMERS, tile)
                cands[f"{t}x{tile}"] = round(busy_timed(ctx, stream, leg.step, reps=7, busy_s=0.1), 4)
            ctx.set_param(cap.PARAM_BLOCK_THREADS, 0)
            ctx.set_param(cap.PARAM_TILE_KMERS, 0)
            again = round(busy_timed(ctx, stream, leg.step, reps=7, busy_s=0.1), 4)  # the launcher's own choice, measured the same way
            name = f"{chosen_shape[0]}x{chosen_shape[1]}" + (" split" if chosen_shape[2] else "")
            best = min(cands, key=cands.get)
            worst = max(cands, key=cands.get)
            shape_report = {"chosen": name, "chosen_ms": again, "candidates_ms": cands, "best": best, "worst": worst,
                            "chosen_over_best": round(again / cands[best], 4),
                            "what": "threads per workgroup x kmers per tile; every candidate forced with KMERS_PARAM_BLOCK_THREADS / _TILE_KMERS into "
                                    "the arrays of the timed leg, 7 launches behind 0.1 s of the same launch each, after the timed region"}
        except Exception as e:  # noqa: BLE001
            shape_report = {"error": repr(e)}
            ctx.set_param(cap.PARAM_BLOCK_THREADS, 0)
            ctx.set_param(cap.PARAM_TILE_KMERS, 0)
    # the two tiny cross-shard reductions of the path, through the same communicator (results known in closed form)
    if comm is not None:
        st, pos, enc = comm.first_error(1 if rank == world - 1 else 0, err_pos=sh.first_base + 5, err_enc=0xF)
        verified &= (st, pos, enc) == (1, plan[-1].first_base + 5, 0xF)
        off, tot = comm.output_offsets(sh.n_kmers)
        verified &= off == sh.first_kmer and tot == sum(s.n_kmers for s in plan)

    fill_gbps = None
    if rank == 0:
        try:  # what this device writes when it does nothing else: torch fills of the two output arrays
            with torch.cuda.stream(stream):
                fills = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    leg.out_k.fill_(1)
                    if leg.out_h is not None:
                        leg.out_h.fill_(2)
                    e1.record(stream)
                    torch.cuda.synchronize()
                    fills.append(e0.elapsed_time(e1))
                fill_bytes = leg.out_k.numel() * 8 + (leg.out_h.numel() * 8 if leg.out_h is not None else 0)
                fill_gbps = fill_bytes / (float(np.median(fills[1:])) * 1e-3) / 1e9
        except Exception as e:
            log(f"fill measurement failed: {e!r}")
    log("[bench] verified, launch shapes and fills done")
    leg.release()

    # ---- weak runs on N > 1 GPUs: the strong split of the north-star input, in the same run -------------------
    strong_extra = None
    strong_bases = args.strong_bases if args.strong_bases >= 0 else (0 if shared_device else NORTH_STAR_BASES)
    if grouped and world > 1 and not strong and strong_bases:
        sleg = Leg(env, strong_bases, GOLDEN ^ 10)
        s_elapsed, s_kern, s_halo = sleg.timed(args.warmup, args.steps)
        s_per_rank = gather_floats(env, [s_elapsed, s_kern, s_halo, float(sleg.sh.n_kmers)])
        s_ok = sleg.verify()
        sleg.release()
        solo = [0.0, 0.0]
        if rank == 0:  # the same input on ONE GPU (rank 0 alone; the others wait at the barrier below)
            env.world, env.grouped, env.comm = 1, False, None
            try:
                one = Leg(env, strong_bases, GOLDEN ^ 10)
                o_elapsed, o_kern, _ = one.timed(args.warmup, args.steps, solo=True)
                s_ok &= one.verify()
                one.release()
                solo = [o_elapsed, o_kern]
            except Exception as e:
                log(f"1-GPU leg of the strong-scaling pair failed: {e!r}")
            env.world, env.grouped, env.comm = world, grouped, comm
        sv = torch.tensor([1 if s_ok else 0], device=dev)
        dist.all_reduce(sv, op=dist.ReduceOp.MIN)
        strong_extra = strong_scaling_entry(args, K, bits, world, strong_bases, s_per_rank, bytes_per_kmer, bool(sv.item()),
                                            solo if rank == 0 else None)
        verified &= bool(sv.item())
    v = torch.tensor([1 if verified else 0], device=dev)
    if grouped:
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
    verified = bool(v.item())

    if rank == 0:
        line = assemble_line(args, K, bits, world, strong, grouped, backend, transport, total_bases, [s.n_bases for s in plan], seed, elapsed,
                             per_rank, bytes_per_kmer, verified, use_pool, write_ceiling, shape_report, plain_alloc, fill_gbps, strong_extra)
        if pool_report is not None:
            line["config"]["pool"] = pool_report
        if fresh_report is not None:
            line["fresh_outputs_per_step"] = fresh_report
        if use_pool:
            ctx.pool_trim()  # (the profiled child processes below make pools of their own)
        rf = line["roofline"]
        # the PMC child passes, rank 0's device (the other ranks wait in the barrier at the end)
        pmc_args = argparse.Namespace(**vars(args))
        if strong:  # the child profiles a launch of the rank's shard size, capped at what fits beside this process's buffers
            pmc_args.bases = max(K, min(sh.n_bases, 2_000_000_000))
        child_timeout = 600 if world == 1 else 200
        log("[bench] line assembled; PMC child passes")
        traffic, source = (None, "not measured (--no-pmc)") if args.no_pmc else measure_traffic(pmc_args, dev_index, child_timeout)
        if traffic is None:
            log(f"roofline.traffic: {source}")
            traffic, source2 = replayed_traffic(args)
            source = f"{source2}; live measurement: {source}"
        rf["traffic"] = traffic
        rf["traffic_source"] = source
        if traffic is not None and pmc_args.bases - K + 1 != rf["kmers_per_launch"]:
            rf["traffic_kmers_per_launch"] = pmc_args.bases - K + 1  # the profiled launch is shorter than the timed one
            rf["traffic_over_algorithmic"] = round(traffic / (bytes_per_kmer * (pmc_args.bases - K + 1)), 4)
        if world == 1 and not args.no_other_configs:
            valu = {} if args.no_pmc else measure_legs(pmc_args, dev_index, child_timeout)
            try:  # informative extras; never allowed to break the headline line
                line["other_configs"] = other_configs(ctx, cap, stream, dev, mem, valu, write_ceiling_gbps=write_ceiling[0],
                                                      write_ceiling_source=write_ceiling[1])
            except Exception as e:
                line["other_configs"] = {"error": repr(e)}
        if not args.no_cpu_baseline:
            log("[bench] other configs done; cpu baseline")
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
