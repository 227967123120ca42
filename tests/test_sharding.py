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
