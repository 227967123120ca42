"""Multi-GPU path on CPU: shard planning + the (K-1)-base halo exchange over torch.distributed
(gloo, world_size 2 and 3).  The per-shard compute is checked with the oracle only as the
CHECKER of the host logic: shards + halo must reproduce exactly the whole-sequence iteration."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_plan_covers_everything_once():
    import kmers_jl_amd
    from kmers_jl_amd.shard import plan_shards
    for bits in (2, 4):
        per_word = 64 // bits
        for k in (1, 3, 31, 33, 64):
            for n_bases in (0, k - 1, k, 100, 4096, 100_003, 10**9, 10**10):
                if n_bases < 0:
                    continue
                for n in (1, 2, 3, 8):
                    plan = plan_shards(n_bases, k, n, bits)
                    n_kmers = max(0, n_bases - k + 1)
                    assert sum(s.n_kmers for s in plan) == n_kmers
                    nxt, nxt_w = 0, 0
                    total_words = (n_bases * bits + 63) // 64
                    for s in plan:
                        assert s.first_kmer == nxt and s.first_word == nxt_w
                        assert s.first_kmer % per_word == 0 or s.n_kmers == 0
                        if s.n_kmers:
                            assert s.first_word == s.first_kmer // per_word
                            need = ((s.first_kmer + s.n_bases) * bits + 63) // 64
                            assert s.first_word + s.n_own_words + s.halo_words >= need
                        nxt += s.n_kmers
                        nxt_w += s.n_own_words
                    assert nxt_w == total_words
                    for g in range(1, n):
                        assert plan[g].send_words == plan[g - 1].halo_words <= plan[g].n_own_words
                    assert plan[-1].halo_words == 0 and plan[0].send_words == 0


def _worker(rank, world, port, k, bits, n_bases, q, transport):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmers_jl_amd  # noqa: F401
        from kmers_jl_amd.shard import HaloExchanger, plan_shards
        from oracle import pyoracle
        orc = pyoracle.get()
        plan = plan_shards(n_bases, k, world, bits)
        sh = plan[rank]
        # every rank generates only the words it owns (the halo is NOT generated locally)
        own = orc.synth_words(4242, sh.first_word, sh.n_own_words, bits)
        buf = torch.zeros(sh.n_own_words + sh.halo_words + 1, dtype=torch.int64)
        buf[:sh.n_own_words] = torch.from_numpy(own.view(np.int64).copy())
        hx = HaloExchanger(buf, sh, plan, transport=transport)
        hx.exchange()
        hx.exchange()  # idempotent: the bench calls it every step
        words = buf.numpy().view(np.uint64)
        kmers, hashes, res = orc.canonical(words, sh.n_bases, bits, 2, k, seed=5)
        assert res.status == 0
        q.put((rank, sh.first_kmer, kmers.copy(), hashes.copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,transport", [(2, "allgather"), (3, "allgather"), (2, "p2p")])
@pytest.mark.parametrize("bits", [2, 4])
def test_halo_exchange_gloo(world, bits, transport):
    from oracle import pyoracle
    orc = pyoracle.get()
    for k, n_bases in ((31, 20_011), (33, 7_000), (3, 999)):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, k, bits, n_bases, q, transport)) for r in range(world)]
        for p in procs:
            p.start()
        parts = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        total_words = (n_bases * bits + 63) // 64
        whole = orc.synth_words(4242, 0, total_words + 1, bits)
        ek, eh, _ = orc.canonical(whole, n_bases, bits, 2, k, seed=5)
        got_k = np.concatenate([p[2] for p in parts])
        got_h = np.concatenate([p[3] for p in parts])
        assert np.array_equal(got_k, ek) and np.array_equal(got_h, eh)


def test_plan_with_stride_and_bytes_matches_c_abi():
    """Stride-lattice + word aligned shards (SpacedKmers, ASCII sources); the Python planner and the
    C ABI's kmers_shard_plan are the same arithmetic."""
    import ctypes as C

    import kmers_jl_amd
    from kmers_jl_amd import _capi
    from kmers_jl_amd.shard import plan_shards

    class CShard(C.Structure):
        _fields_ = [(n, C.c_uint64) for n in ("first_kmer", "n_kmers", "first_base", "n_bases", "first_word",
                                              "n_own_words")] + [("halo_words", C.c_uint32), ("send_words", C.c_uint32)]
    lib = _capi.load()
    for bits in (2, 4, 8):
        per_word = 64 // bits
        for k in (1, 3, 21, 31, 64):
            for stride in (1, 2, 3, 7, 16, 21, 40, 100):
                for n_bases in (0, k - 1, k, 777, 100_003, 10**9 + 7):
                    for n in (1, 2, 3, 8):
                        plan = plan_shards(n_bases, k, n, bits, stride)
                        m = (n_bases - k) // stride + 1 if n_bases >= k else 0
                        assert sum(s.n_kmers for s in plan) == m
                        nxt, nxt_w = 0, 0
                        for g, s in enumerate(plan):
                            cs = CShard()
                            assert lib.kmers_shard_plan(n_bases, k, stride, bits, n, g, C.byref(cs)) == 0
                            assert (cs.first_kmer, cs.n_kmers, cs.first_base, cs.n_bases, cs.first_word, cs.n_own_words,
                                    cs.halo_words, cs.send_words) == \
                                (s.first_kmer, s.n_kmers, s.first_base, s.n_bases, s.first_word, s.n_own_words,
                                 s.halo_words, s.send_words), (bits, k, stride, n_bases, n, g)
                            assert s.first_kmer == nxt and s.first_word == nxt_w
                            if s.n_kmers:
                                assert s.first_base % per_word == 0 and s.first_base % stride == 0
                                assert s.first_word == s.first_base // per_word
                                need = ((s.first_base + s.n_bases) * bits + 63) // 64
                                assert s.first_word + s.n_own_words + s.halo_words >= need
                            nxt += s.n_kmers
                            nxt_w += s.n_own_words
                        assert nxt_w == (n_bases * bits + 63) // 64
                        for g in range(1, n):
                            assert plan[g].send_words == plan[g - 1].halo_words <= plan[g].n_own_words
    bad = CShard()
    assert lib.kmers_shard_plan(100, 0, 1, 2, 2, 0, C.byref(bad)) != 0
    assert lib.kmers_shard_plan(100, 3, 1, 3, 2, 0, C.byref(bad)) != 0
    assert lib.kmers_shard_plan(100, 3, 1, 2, 2, 2, C.byref(bad)) != 0


def _worker_iterators(rank, world, port, n_bases, ambig, q):
    """Spaced (strict, error = min over shards) and Unambiguous (counts scanned over shards) on shards."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import kmers_jl_amd  # noqa: F401
        from kmers_jl_amd.shard import HaloExchanger, first_error, output_offsets, plan_shards
        from oracle import pyoracle
        orc = pyoracle.get()
        out = {}
        for name, k, stride in (("spaced", 21, 3), ("spaced_wide", 5, 40), ("unambiguous", 21, 1)):
            plan = plan_shards(n_bases, k, world, 4, stride)
            sh = plan[rank]
            own = orc.synth_words(99, sh.first_word, sh.n_own_words, 4, ambig)
            buf = torch.zeros(sh.n_own_words + sh.halo_words + 1, dtype=torch.int64)
            buf[:sh.n_own_words] = torch.from_numpy(own.view(np.int64).copy())
            HaloExchanger(buf, sh, plan).exchange()
            words = buf.numpy().view(np.uint64)
            if name == "unambiguous":
                km, st, res = orc.unambiguous(words, sh.n_bases, 4, k)
                off, total = output_offsets(len(km))
                out[name] = (off, total, km.copy(), st + sh.first_base)  # index_origin = first_base
            else:
                km, res = orc.spaced(words, sh.n_bases, 4, 2, k, stride)
                err = first_error(res.status, res.err_pos + sh.first_base if res.status == 1 else 0, res.err_enc)
                out[name] = (err, km.copy() if res.status == 0 else None)
        q.put((rank, out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("ambig", [0, 40])
def test_sharded_spaced_and_unambiguous_gloo(world, ambig):
    from oracle import pyoracle
    orc = pyoracle.get()
    n_bases = 30_011
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=_worker_iterators, args=(r, world, port, n_bases, ambig, q)) for r in range(world)]
    for p in procs:
        p.start()
    parts = [o for _, o in sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole = orc.synth_words(99, 0, (n_bases * 4 + 63) // 64 + 1, 4, ambig)
    for name, k, stride in (("spaced", 21, 3), ("spaced_wide", 5, 40)):
        ek, eres = orc.spaced(whole, n_bases, 4, 2, k, stride)
        for o in parts:  # every rank holds the same, global, first error
            assert o[name][0] == ((1, eres.err_pos, eres.err_enc) if eres.status == 1 else (0, 0, 0)), name
        if eres.status == 0:
            assert np.array_equal(np.concatenate([o[name][1] for o in parts]), ek)
    assert ambig == 0 or orc.spaced(whole, n_bases, 4, 2, 21, 3)[1].status == 1
    ek, es, _ = orc.unambiguous(whole, n_bases, 4, 21)
    offs = [o["unambiguous"][0] for o in parts]
    assert all(o["unambiguous"][1] == len(ek) for o in parts)
    assert offs == list(np.cumsum([0] + [len(o["unambiguous"][2]) for o in parts])[:-1])
    assert np.array_equal(np.concatenate([o["unambiguous"][2] for o in parts]), ek)
    assert np.array_equal(np.concatenate([o["unambiguous"][3] for o in parts]), es)
