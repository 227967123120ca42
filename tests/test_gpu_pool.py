"""The device's class pool (csrc/pool_api.hip) on hardware: the virtual-memory calls it rests on behave as the code assumes,
blocks are assembled by region class (the arrays of a launch differ at every position, a lone output's halves differ), data
survives, and -- round 6 -- what a block COSTS: a freed block stays mapped and the next request of its shape takes it back
without a call into the driver, kmers_dev_free does not wait, the next user of a block comes after the work that was queued
when it was freed, the hoard is bounded, exhaustion is an error that leaves the pool consistent, host threads and a second
process may hammer it (reference: Base.collect makes a fresh Vector per call over src/iterators/CanonicalKmers.jl:199-225;
src/kmer.jl:255-261)."""
import ctypes as C
import gc
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# torch FIRST: it brings its own copy of the HIP / HSA runtime libraries, and a process that has already loaded /opt/rocm's
# (through libkmers_hip.so) cannot initialise torch's afterwards ("No HIP GPUs are available")
torch = pytest.importorskip("torch")

import kmers_jl_amd as km  # noqa: E402
from kmers_jl_amd import _capi as cap  # noqa: E402

MiB, GiB = 1 << 20, 1 << 30
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX = np.uint64(0x517cc1b727220a95)


@pytest.fixture()
def ctx():
    gc.collect()   # (device arrays that earlier test modules left to the collector are blocks that are OUT: a new block would be placed beside them)
    c = km.Context(0)
    yield c
    c.close()


def h2d(ctx, ptr, arr):
    ctx.check(ctx.lib.kmers_memcpy_h2d(ctx.handle, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")


def d2h(ctx, ptr, n):
    back = np.zeros(n, dtype=np.uint64)
    ctx.check(ctx.lib.kmers_memcpy_d2h(ctx.handle, back.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), back.nbytes), "d2h")
    return back


def hoard_ok(st):
    """include/kmers_hip.h: outside its blocks (out or cached) the pool holds at most max(4 GiB, a quarter of them)."""
    blocks = st["in_use"] + st["cached"]
    return st["held"] <= blocks + max(4 * GiB, blocks // 4), st


def test_vmm_calls_behave_as_the_pool_assumes(ctx):
    # KMERS_OK = fresh mappings show their memory and a re-used range shows its NEW memory after the pool's flush; the flag says
    # whether the stale-translation behaviour the flush exists for was reproduced (it is on ROCm 7.2; either answer is fine)
    assert ctx.pool_selftest() in (True, False)
    ctx.pool_trim()
    assert ctx.pool_info()["held"] == 0


def test_blocks_are_assembled_by_class_and_hold_their_data(ctx):
    na, nb = 6 * GiB + 5 * MiB, 6 * GiB
    a = ctx.alloc(na)
    b = ctx.alloc(nb)
    chunk, ca = ctx.pool_layout(a)
    _, cb = ctx.pool_layout(b)
    assert chunk == GiB and len(ca) == 7 and len(cb) == 6
    info = ctx.pool_info()
    assert info["in_use"] == 13 * GiB and info["held"] >= info["in_use"] + info["n_classes"] * GiB
    assert hoard_ok(ctx.pool_stats())[0], ctx.pool_stats()   # what the search walked past has gone back to the driver
    assert 2 <= info["n_classes"] <= 4, info     # every MI355X so far has shown its three classes within the first gigabytes
    assert info["two_class_gbps"] > 1.08 * info["one_class_gbps"] > 5000, info
    # the arrays of one launch: different classes at every relative position
    assert all(ca[int((i + 0.5) * GiB / nb * na) // GiB] != cb[i] for i in range(6)), (ca, cb)   # (relative BYTE positions of the two arrays)
    # ... and that is what the device does with them (kmers_placement_probe: two store streams side by side, DESTRUCTIVE)
    apart, together = ctx.placement_probe(a, b, GiB), ctx.placement_probe(a, a + 2 * GiB, GiB) if ca[0] == ca[2] else 0
    assert apart > 6700, (apart, ca, cb)
    if together:
        assert together < 0.95 * apart, (apart, together)
    # a lone output: second half against first half
    lone = ctx.alloc(8 * GiB, lone_output=True)
    cl = ctx.pool_layout(lone)[1]
    assert all(cl[i] != cl[i + 4] for i in range(4)), cl
    assert ctx.placement_probe(lone, lone + 4 * GiB, GiB) > 6700
    # ... whatever the size: the array's MIDDLE lies on a handle boundary (the block is the two halves rounded up, the pointer inside it)
    odd_bytes = 10_000_000_000 - 8 * 30          # C3's output: 9.31 GiB
    odd = ctx.alloc(odd_bytes, lone_output=True)
    co = ctx.pool_layout(odd)[1]
    assert len(co) == 10 and all(co[i] != co[i + 5] for i in range(5)), co     # (position by position: a half may be a mix, [B B B B A | A A A A C])
    half = -(-(odd_bytes // 2) // 4096) * 4096                 # the second half begins on the boundary between handle 4 and handle 5
    assert odd % 4096 == 0 and ctx.placement_probe(odd, odd + half, GiB) > 6700 and ctx.placement_probe(odd + half - GiB, odd + 2 * half - GiB, GiB) > 6700
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(odd + 4096)) == cap.E_BADARG
    ctx.free(odd)
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(odd)) == cap.E_BADARG     # freed twice: an error, not a corruption
    # a pattern across every chunk boundary, written and read through the C ABI's copies
    rng = np.random.default_rng(11)
    for off in (0, GiB - 4096, 3 * GiB - 8, na - 8192):
        data = rng.integers(0, 1 << 63, 1024, dtype=np.uint64)
        h2d(ctx, a + off, data)
        assert np.array_equal(data, d2h(ctx, a + off, 1024))
    for p in (a, b, lone):
        ctx.free(p)
    st = ctx.pool_stats()
    assert st["in_use"] == 0 and st["blocks_out"] == 0 and st["cached_blocks"] >= 3 and hoard_ok(st)[0], st
    assert ctx.pool_trim() == st["held"] and ctx.pool_info()["held"] == 0


def test_arrays_below_a_gigabyte_are_one_handle_each_in_different_classes(ctx):
    """128 MiB to 1 GiB (a 30-60 Mbase chromosome: 0.25-0.5 GB per output array): one whole handle of a chosen class, addressed
    through the handle's own mapping -- nothing is mapped or unmapped for it, ever."""
    p = ctx.alloc(cap.POOL_MIN_BYTES - 1)
    assert ctx.pool_layout(p)[1] == []            # smaller: a plain allocation
    ctx.free(p)
    a, b = ctx.alloc(300 * MiB), ctx.alloc(300 * MiB)
    la, lb = ctx.pool_layout(a)[1], ctx.pool_layout(b)[1]
    assert len(la) == len(lb) == 1 and la[0] != lb[0], (la, lb)
    assert ctx.placement_probe(a, b, 256 * MiB) > 6500
    w = ctx.alloc(200 * MiB)                       # a third array (the sequence): not in the class of the array before it
    lw = ctx.pool_layout(w)[1]                     # (and in the third class if the pool has a free handle of it)
    assert len(lw) == 1 and lw[0] != lb[0], (la, lb, lw)
    rng = np.random.default_rng(3)
    data = rng.integers(0, 1 << 63, 4096, dtype=np.uint64)
    h2d(ctx, a + 300 * MiB - data.nbytes, data)
    assert np.array_equal(d2h(ctx, a + 300 * MiB - data.nbytes, 4096), data)
    before = ctx.pool_stats()
    for p in (a, b, w):
        ctx.free(p)
    for _ in range(5):                              # a loop over such arrays: served from the cache, no handle created
        x, y = ctx.alloc(300 * MiB), ctx.alloc(300 * MiB)
        assert ctx.pool_layout(x)[1] != ctx.pool_layout(y)[1]
        ctx.free(x)
        ctx.free(y)
    after = ctx.pool_stats()
    assert after["chunks_created"] == before["chunks_created"] and after["cache_hits"] == before["cache_hits"] + 10, (before, after)
    ctx.set_param(cap.PARAM_POOL, 0)
    q = ctx.alloc(2 * GiB)
    assert ctx.pool_layout(q)[1] == []
    ctx.free(q)
    ctx.set_param(cap.PARAM_POOL, 1)
    r = ctx.alloc(GiB)
    assert len(ctx.pool_layout(r)[1]) == 1
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(r + 4096)) == cap.E_BADARG   # not the start of the block
    ctx.free(r)


def test_alloc_free_loop_reuses_memory_and_never_shows_stale_data(ctx):
    # the reference's collect: one array per call.  Sizes change from call to call so that chunks move between address ranges;
    # what a block shows must always be what was last written through IT.
    rng = np.random.default_rng(5)
    for it in range(12):
        sizes = [int(rng.integers(1, 4)) * GiB + int(rng.integers(0, 2)) * 4096 for _ in range(3)]
        blocks = [ctx.alloc(s) for s in sizes]
        tags = []
        for b, s in zip(blocks, sizes):
            t = rng.integers(0, 1 << 63, 2, dtype=np.uint64)
            tags.append(t)
            h2d(ctx, b, t[:1])
            h2d(ctx, b + s - 8, t[1:])
        for b, s, t in zip(blocks, sizes, tags):
            assert d2h(ctx, b, 1)[0] == t[0] and d2h(ctx, b + s - 8, 1)[0] == t[1], (it, hex(b))
        for b in blocks[::-1] if it % 2 else blocks:
            ctx.free(b)
        st = ctx.pool_stats()
        assert st["in_use"] == 0 and hoard_ok(st)[0], (it, st)
    st = ctx.pool_stats()
    assert st["cache_hits"] > 0 and st["held"] <= 5 * 12 * GiB, st
    released = ctx.pool_trim()
    assert released == st["held"] and ctx.pool_info()["held"] == 0
    # with the cache switched off a free takes the block apart at once (and waits for the device's contexts)
    ctx.set_param(cap.PARAM_POOL_CACHE, 0)
    p = ctx.alloc(2 * GiB)
    ctx.free(p)
    st = ctx.pool_stats()
    assert st["cached"] == 0 and st["cached_blocks"] == 0 and hoard_ok(st)[0], st
    # ... and every allocation is a NEW mapping: none of them begins on a GiB boundary (a 1 GiB handle mapped on one is a single 1 GiB
    # page, and under the pool's churn of handles the device now and then faults at its first access: csrc/pool_api.hip, reserve();
    # tools/device_probes/vmm_churn.hip; the runtime hands out such an address about once in thirty reservations)
    for it in range(40):
        size = (GiB // 2, GiB, 2 * GiB, 3 * GiB)[it % 4]
        p = ctx.alloc(size)
        assert ctx.pool_layout(p)[1] and p % GiB != 0, hex(p)
        t = rng.integers(0, 1 << 63, 1, dtype=np.uint64)
        h2d(ctx, p + size - 8, t)
        assert d2h(ctx, p + size - 8, 1)[0] == t[0]
        ctx.free(p)
    ctx.set_param(cap.PARAM_POOL_CACHE, 1)


def test_pool_is_shared_by_the_contexts_of_a_device_and_outlives_the_first():
    a, b = km.Context(0), km.Context(0)
    pa = a.alloc(GiB)
    pb = b.alloc(2 * GiB)
    assert a.pool_info()["in_use"] == b.pool_info()["in_use"] == 3 * GiB
    assert a.pool_layout(pa)[1][0] != b.pool_layout(pb)[1][0]   # the second block beside the first, whoever asked
    b.free(pa)  # any context of the device may free a block
    a.close()   # the pool stays: b still uses it
    assert b.pool_info()["in_use"] == 2 * GiB and b.pool_layout(pb)[1]
    b.free(pb)
    b.close()
    c = km.Context(0)
    assert c.pool_info()["held"] == 0  # the last context out returned everything
    c.close()


def canonical_launch(ctx, d_words, L, K, d_k, d_h, flags):
    seq = cap.Seq(d_words, L, 0, 0, 4, 0)
    res = cap.Result()
    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, d_k, d_h, 0, flags, C.byref(res))
    assert rc == 0, ctx.last_error()
    return res


@pytest.mark.parametrize("mbases", [48, 320])
def test_collect_loop_with_fresh_outputs_costs_microseconds(ctx, orc, mbases):
    """{alloc kmers, alloc hashes, launch, free, free} -- what `collect(CanonicalDNAMers{31}(seq))` per sequence amounts to -- beside
    the same launches into resident arrays: after the first round every block comes from the pool's cache (no handle created, no
    block assembled), a free does not wait for the launch, and the loop costs microseconds per call more than the resident one.
    48 Mbase: arrays of one handle each (their home mappings); 320 Mbase: three handles each (mapped blocks)."""
    L, K = mbases * 1_000_000 + 37, 31
    nw, n = (L * 4 + 63) // 64, L - K + 1
    d_words = ctx.alloc(nw * 8 + 8)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 21, 0, nw, 4, 0, d_words), "kmers_synth_dna")
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    rounds = 30
    # resident arrays
    d_k, d_h = ctx.alloc(n * 8), ctx.alloc(n * 8)
    assert ctx.pool_layout(d_k)[1] and all(x != y for x, y in zip(ctx.pool_layout(d_k)[1], ctx.pool_layout(d_h)[1]))
    for _ in range(3):
        canonical_launch(ctx, d_words, L, K, d_k, d_h, ASYNC)
    assert ctx.sync()[0] == 0
    t0 = time.perf_counter()
    for _ in range(rounds):
        canonical_launch(ctx, d_words, L, K, d_k, d_h, ASYNC)
    assert ctx.sync()[0] == 0
    resident = (time.perf_counter() - t0) / rounds
    expect_k = d2h(ctx, d_k, 4096)
    ctx.free(d_k)
    ctx.free(d_h)
    # fresh arrays per launch
    before = ctx.pool_stats()
    alloc_us, free_us = [], []
    t0 = time.perf_counter()
    for r in range(rounds):
        t = time.perf_counter()
        d_k, d_h = ctx.alloc(n * 8), ctx.alloc(n * 8)
        alloc_us.append((time.perf_counter() - t) * 1e6 / 2)
        canonical_launch(ctx, d_words, L, K, d_k, d_h, ASYNC)
        if r + 1 < rounds:
            t = time.perf_counter()
            ctx.free(d_k)
            ctx.free(d_h)
            free_us.append((time.perf_counter() - t) * 1e6 / 2)
    assert ctx.sync()[0] == 0
    fresh = (time.perf_counter() - t0) / rounds
    after = ctx.pool_stats()
    assert after["cache_hits"] - before["cache_hits"] == 2 * rounds, (before, after)            # every block from the cache
    assert after["chunks_created"] == before["chunks_created"] and after["evictions"] == before["evictions"], (before, after)
    assert all(x != y for x, y in zip(ctx.pool_layout(d_k)[1], ctx.pool_layout(d_h)[1]))        # ... and still in different classes
    kmers, hashes = d2h(ctx, d_k, n), d2h(ctx, d_h, n)
    assert np.array_equal(kmers[:4096], expect_k) and np.array_equal(hashes, kmers * FX)
    words = orc.synth_words(21, 0, nw, 4)
    ek, eh, _ = orc.canonical(words, 200_000 + K - 1, 4, 2, K)
    assert np.array_equal(kmers[:200_000], ek[:, 0]) and np.array_equal(hashes[:200_000], eh)
    print(f"\n{mbases} Mbase: resident {resident * 1e3:.4f} ms per launch, fresh outputs {fresh * 1e3:.4f} ms; alloc {np.median(alloc_us):.1f} us, "
          f"free {np.median(free_us):.1f} us (medians per call, through ctypes)")
    assert np.median(alloc_us) < 200 and np.median(free_us) < 200, (np.median(alloc_us), np.median(free_us))
    assert fresh <= 1.05 * resident + 100e-6, (fresh, resident)   # (through Python; bench.py's leg is the measurement)
    for p in (d_k, d_h, d_words):
        ctx.free(p)


def test_the_next_user_of_a_freed_block_comes_after_the_work_queued_before_the_free(orc):
    """kmers_dev_free does not wait.  Context A launches 1 Gbase into a block and frees it while the launch runs; context B (another
    stream) asks for the same shape, gets the SAME block from the cache and writes the block's tail with a short launch of its own.
    If B's launch did not wait for A's on the device, A -- which reaches the tail two milliseconds later -- would overwrite it."""
    a, b = km.Context(0), km.Context(0)
    L, K = 1_000_000_000, 31
    nw, n = (L * 4 + 63) // 64, L - K + 1
    wa = a.alloc(nw * 8 + 8)
    a.check(a.lib.kmers_synth_dna(a.handle, 31, 0, nw, 4, 0, wa), "kmers_synth_dna")
    Ls = 2_000_003
    nws, ns = (Ls * 4 + 63) // 64, Ls - K + 1
    small = orc.synth_words(77, 0, nws, 4)
    wb = b.alloc(256 * MiB)
    b.h2d(wb, small)
    ek, _, _ = orc.canonical(small, Ls, 4, 2, K)
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    for trial in range(3):
        blk = a.alloc(n * 8)
        canonical_launch(a, wa, L, K, blk, None, ASYNC)      # about 1.3 ms of stores, the tail last
        a.free(blk)                                          # ... while it runs
        again = b.alloc(n * 8)
        assert again == blk, "the cache did not hand the freed block to the next request of its shape"
        tail = again + (n - ns) * 8 // 16 * 16
        canonical_launch(b, wb, Ls, K, tail, None, ASYNC)
        assert b.sync()[0] == 0 and a.sync()[0] == 0
        got = d2h(b, tail, ns)
        assert np.array_equal(got, ek[:, 0]), f"trial {trial}: the earlier launch of the block's previous owner overwrote its next user's data"
        b.free(again)
    a.free(wa)
    b.free(wb)
    a.close()
    b.close()


def test_exhaustion_is_an_error_that_leaves_the_pool_consistent(ctx):
    free_b, total_b = torch.cuda.mem_get_info(0)
    assert free_b > 64 * GiB, "this test needs a mostly empty device"
    big = int(free_b * 0.6) // GiB * GiB
    a = ctx.alloc(big)
    t0 = time.perf_counter()
    p = C.c_void_p()
    rc = ctx.lib.kmers_dev_alloc(ctx.handle, big, C.byref(p))
    assert rc == cap.E_NOMEM and not p.value, (rc, ctx.last_error())
    assert time.perf_counter() - t0 < 5.0, "a request that cannot fit must not walk the device first"
    st = ctx.pool_stats()
    assert st["in_use"] == big and st["blocks_out"] == 1 and hoard_ok(st)[0], st
    small = ctx.alloc(2 * GiB)                      # the pool still works
    assert len(ctx.pool_layout(small)[1]) == 2
    ctx.free(small)
    ctx.free(a)
    b = ctx.alloc(big)                              # ... and the memory of the first block serves the request that failed
    assert ctx.pool_stats()["in_use"] == big
    ctx.free(b)
    rc = ctx.lib.kmers_dev_alloc(ctx.handle, 4 * total_b, C.byref(p))
    assert rc == cap.E_NOMEM and not p.value
    assert ctx.pool_trim() > 0 and ctx.pool_info()["held"] == 0
    # a capped pool: what it cannot hold is a plain allocation, not an error (the cap holds for the calibration's handles too)
    ctx.set_param(cap.PARAM_POOL_MAX_GIB, 3)
    q = ctx.alloc(2 * GiB)
    assert ctx.pool_layout(q)[1] == [] and ctx.pool_info()["held"] == 0
    ctx.free(q)
    ctx.set_param(cap.PARAM_POOL_MAX_GIB, 8)
    blocks, from_pool = [], []
    for _ in range(3):                               # (two or three yardsticks + 3 x 3 GiB > 8: the last one cannot come from the pool)
        blocks.append(ctx.alloc(3 * GiB))
        from_pool.append(len(ctx.pool_layout(blocks[-1])[1]) == 3)
        assert ctx.pool_info()["held"] <= 8 * GiB, ctx.pool_stats()
    assert from_pool[0] and not from_pool[-1], from_pool
    data = np.arange(1024, dtype=np.uint64)
    for b in blocks:
        h2d(ctx, b + 3 * GiB - data.nbytes, data)
        assert np.array_equal(d2h(ctx, b + 3 * GiB - data.nbytes, 1024), data)
    for b in blocks:
        ctx.free(b)
    ctx.set_param(cap.PARAM_POOL_MAX_GIB, 0)


def test_four_threads_allocate_launch_and_free_at_once(orc):
    """One pool per device behind one mutex: four host threads, a context each, allocate blocks of changing sizes, fill them with a
    launch, free them in any order -- every block must hold what was written through it, the books must balance at the end."""
    errors = []
    n_threads, rounds = 4, 20
    expect = {}

    def head(seed, bits):
        if (seed, bits) not in expect:
            expect[(seed, bits)] = orc.synth_words(seed, 0, 4096, bits)
        return expect[(seed, bits)]
    for t in range(n_threads):
        for r in range(rounds):
            for j in range(3):
                head(1000 * t + 10 * r + j, 4)

    def worker(tid):
        try:
            c = km.Context(0)
            rng = np.random.default_rng(100 + tid)
            for r in range(rounds):
                sizes = [int(rng.choice([160 * MiB, 700 * MiB, GiB + 4096, 2 * GiB, 3 * GiB - 8192])) for _ in range(3)]
                blocks = [c.alloc(s, lone_output=bool(rng.integers(0, 2)) and s >= 2 * GiB) for s in sizes]
                for j, (p, s) in enumerate(zip(blocks, sizes)):
                    nw = s // 8
                    c.check(c.lib.kmers_synth_dna(c.handle, 1000 * tid + 10 * r + j, 0, nw, 4, 0, p), "kmers_synth_dna")
                for j in rng.permutation(3):
                    p, s = blocks[j], sizes[j]
                    want = head(1000 * tid + 10 * r + int(j), 4)
                    got = np.zeros(4096, np.uint64)
                    c.d2h(got, p)
                    assert np.array_equal(got, want), (tid, r, int(j), hex(p))
                    c.free(p)
            c.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))

    keeper = km.Context(0)   # (keeps the pool alive while the workers come and go)
    keeper.alloc(GiB)
    threads = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=900)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
    st = keeper.pool_stats()
    assert st["in_use"] == GiB and st["blocks_out"] == 1 and hoard_ok(st)[0], st
    keeper.close()


SECOND_PROCESS = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import kmers_jl_amd as km
from kmers_jl_amd import _capi as cap
ctx = km.Context(0)
L, K = 150_000_011, 31
nw, n = (L * 4 + 63) // 64, L - K + 1
for it in range(6):
    w, k, h = ctx.alloc(nw * 8 + 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
    assert len(ctx.pool_layout(k)[1]) == 2 and len(ctx.pool_layout(h)[1]) == 2
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 5 + it, 0, nw, 4, 0, w), "synth")
    seq, res = cap.Seq(w, L, 0, 0, 4, 0), cap.Result()
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, k, h, 0, cap.MEM_DEVICE, C.byref(res)) == 0 and res.n_out == n
    kk, hh = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    ctx.d2h(kk, k); ctx.d2h(hh, h)
    assert np.array_equal(hh, kk * np.uint64(0x517cc1b727220a95)) and int((kk >> np.uint64(62)).max()) == 0
    x = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(x), cap.MEM_DEVICE, C.byref(res)) == 0
    assert int(np.bitwise_xor.reduce(kk)) == x.value
    for p in (w, k, h):
        ctx.free(p)
st = ctx.pool_stats()
assert st["in_use"] == 0 and st["cache_hits"] >= 10, st
ctx.close()
print("second process ok")
"""


def test_two_processes_use_the_device_at_once(tmp_path):
    """Every process has a pool of its own (physical handles are a process's): two of them allocate, classify (their probes run
    beside each other's launches), launch and free on one device at the same time; both must get the right elements."""
    script = tmp_path / "second.py"
    script.write_text(SECOND_PROCESS)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for _ in range(2)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "second process ok" in o, o[-3000:]


def test_launches_into_pool_blocks_match_the_oracle(ctx, orc):
    # 140 Mbase: both outputs span two chunks, so the launch crosses a chunk boundary of each
    L, K, bits = 140_000_037, 31, 4
    nw = (L * bits + 63) // 64
    n = L - K + 1
    d_words = ctx.alloc(nw * 8 + 8)
    d_k, d_h = ctx.alloc(n * 8), ctx.alloc(n * 8)
    lk, lh = ctx.pool_layout(d_k)[1], ctx.pool_layout(d_h)[1]
    assert len(lk) == len(lh) == 2 and all(x != y for x, y in zip(lk, lh)), (lk, lh)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 13, 0, nw, bits, 0, d_words), "kmers_synth_dna")
    seq = cap.Seq(d_words, L, 0, 0, bits, 0)
    res = cap.Result()
    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, d_k, d_h, 0, cap.MEM_DEVICE, C.byref(res))
    assert rc == 0 and res.n_out == n, ctx.last_error()
    assert ctx.last_launch_shape()[:2] == (128, 1536)   # the shape for two well-placed one-word arrays (stream_launch.hpp)
    kmers, hashes = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    ctx.d2h(kmers, d_k)
    ctx.d2h(hashes, d_h)
    words = orc.synth_words(13, 0, nw, bits)

    def window(lo, count):  # the oracle over [lo, lo + count): its views start at symbol 0 of a word
        w0 = lo * bits // 64
        first = lo - w0 * (64 // bits)
        ek, eh, eres = orc.canonical(words[w0:], first + count + K - 1, bits, 2, K)
        assert eres.status == 0
        return ek[first:first + count, 0], eh[first:first + count]

    for lo, count in ((0, 2_000_000), (GiB // 8 - 5000, 10_000), (n - 100_000, 100_000)):   # head, the chunk boundary, tail
        ek, eh = window(lo, count)
        assert np.array_equal(kmers[lo:lo + count], ek) and np.array_equal(hashes[lo:lo + count], eh), lo
    assert np.array_equal(hashes, kmers * FX)   # every element: fx_hash of one word, seed 0 (src/kmer.jl:255-261)
    # the lone output of a launch: a block whose second half lies in another class than its first (here: two handles, the array
    # placed so that its middle is their boundary), written in split order; same elements
    blk = ctx.alloc(2 * GiB, lone_output=True)
    cl = ctx.pool_layout(blk)[1]
    assert len(cl) == 2 and cl[0] != cl[1], cl
    d_l = blk + GiB - (n * 4) // 16 * 16
    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, d_l, None, 0, cap.MEM_DEVICE, C.byref(res))
    assert rc == 0 and res.n_out == n, ctx.last_error()
    assert ctx.last_launch_shape()[2] == 1   # two write windows
    lone = np.zeros(n, np.uint64)
    ctx.d2h(lone, d_l)
    assert np.array_equal(lone, kmers)
    for p in (d_words, d_k, d_h, blk):
        ctx.free(p)


def shifted_view(words, first, bits):
    """the oracle takes a view that starts at symbol 0: the words of a view that starts `first` symbols in"""
    shifted = np.zeros(len(words), np.uint64)
    sh = first * bits
    w0, b0 = sh // 64, sh % 64
    src64 = words[w0:]
    shifted[:len(src64)] = src64 >> np.uint64(b0)
    if b0:
        shifted[:len(src64) - 1] |= src64[1:] << np.uint64(64 - b0)
    return shifted


def test_lone_output_launches_fuzzed(ctx, orc):
    """Random geometries through the lone-output path (a block taken by role: its halves in two classes, split order, its own launch
    shapes): kmer widths of one to four words, both kmer alphabets, strides, views that start anywhere in a word, odd lengths --
    every element against the oracle.  The array is laid across the middle of a two-handle block, so that the launcher finds its
    halves in two classes at sizes the oracle covers in seconds."""
    rng = np.random.default_rng(2024)
    res = cap.Result()
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    split_seen = 0
    for case in range(14):
        bits = int(rng.choice([2, 4]))
        dst = int(rng.choice([2, 2, 4]))
        K = int(rng.choice([1, 5, 21, 31, 32, 33, 63, 64])) if dst == 2 else int(rng.choice([3, 16, 17, 31, 32]))
        what = str(rng.choice(["canonical", "fw", "spaced", "tuples"]))
        J = int(rng.choice([2, 3, 5, 16])) if what == "spaced" else 1
        N = (K * dst + 63) // 64
        per = 2 * N if what == "tuples" else N
        n_min = (64 << 20) // (8 * per) + 1000
        L = (n_min + int(rng.integers(0, 300_000))) * J + K
        first = int(rng.choice([0, 1, 7, 15, 16, 31, 33]))
        words = orc.synth_words(case, 0, ((first + L) * bits + 63) // 64 + 1, bits)
        d_w = ctx.alloc(max(words.nbytes, 128 * MiB))
        ctx.h2d(d_w, words)
        seq = cap.Seq(d_w, L, first, 0, bits, 0)
        shifted = shifted_view(words, first, bits)
        if what == "canonical":
            exp = orc.canonical(shifted, L, bits, dst, K)[0]
            call = lambda out: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, out, None, 0, ASYNC, C.byref(res))
        elif what == "fw":
            exp = orc.fw_kmers(shifted, L, bits, dst, K)[0]
            call = lambda out: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, out, None, ASYNC, C.byref(res))
        elif what == "spaced":
            exp = orc.spaced(shifted, L, bits, dst, K, J)[0]
            call = lambda out: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, out, ASYNC, C.byref(res))
        else:
            f, r, _ = orc.fwrv(shifted, L, bits, dst, K)
            exp = np.concatenate([f, r], axis=1)
            call = lambda out: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, out, None, ASYNC | cap.OUT_TUPLES, C.byref(res))
        assert exp.nbytes >= 64 << 20
        blk = ctx.alloc(2 * GiB, lone_output=True)
        cl = ctx.pool_layout(blk)[1]
        out = blk + GiB - (exp.nbytes // 2) // 16 * 16         # the array's middle on the boundary of the block's two handles
        assert call(out) == 0, ctx.last_error()
        assert ctx.sync()[0] == 0
        split_seen += ctx.last_launch_shape()[2] if cl[0] != cl[1] else 1
        host = np.zeros(exp.shape, np.uint64)
        ctx.d2h(host, out)
        ctx.free(blk)
        ctx.free(d_w)
        assert np.array_equal(host, exp), (case, bits, dst, K, what, J, L, first)
    assert split_seen >= 10, "the lone-output launches of this test were not written through two windows"


def test_two_output_launches_placed_by_the_pool_and_shaped_by_the_launcher(ctx, orc):
    """The launch shapes of the headline path (csrc/stream_launch.hpp: 128 x 1536 for two one-word arrays, 256 x 768 for two-word
    kmers + reverse complements, 128 x 768 for two-word canonical kmers + hashes) are picked by the LAUNCHER when it sees its two
    output arrays in blocks of the pool that differ in class.  Nothing is forced: kmers_dev_alloc for every buffer, the library's
    own choice (kmers_last_launch_shape says what it was) -- and EVERY element of 128 MiB and more per array against the oracle:
    C2 (CanonicalDNAMers{31} + fx_hash), C4 (FwDNAMers{63} + reverse complements), two-word canonical kmers + hashes
    (src/iterators/CanonicalKmers.jl:131-144, :220-225; src/kmer.jl:255-261)."""
    res = cap.Result()
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    seen = {}
    for what, K, L in (("c2", 31, 17_000_031), ("c4", 63, 8_500_063), ("canon2", 63, 17_000_063)):
        words = orc.synth_words(41 + K, 0, (L * 4 + 63) // 64 + 1, 4)
        n = L - K + 1
        N = (2 * K + 63) // 64
        out_a = ctx.alloc(8 * n * N)       # the two outputs one after the other: the pool puts them into two classes
        out_b = ctx.alloc(8 * n * (N if what == "c4" else 1))
        assert ctx.pool_layout(out_a)[1] and ctx.pool_layout(out_b)[1], "both arrays must be blocks of the pool for this test"
        d_w = ctx.alloc(max(words.nbytes, 128 * MiB))
        ctx.h2d(d_w, words)
        seq = cap.Seq(d_w, L, 0, 0, 4, 0)
        if what == "c4":
            rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, out_a, out_b, ASYNC, C.byref(res))
            ea, eb, _ = orc.fwrv(words, L, 4, 2, K)
        else:
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, out_a, out_b, 5, ASYNC, C.byref(res))
            ea, eb, _ = orc.canonical(words, L, 4, 2, K, seed=5)
        assert rc == 0 and ctx.sync()[0] == 0, ctx.last_error()
        seen[what] = ctx.last_launch_shape()
        ga, gb = np.zeros(ea.shape, np.uint64), np.zeros(eb.shape, np.uint64)
        ctx.d2h(ga, out_a)
        ctx.d2h(gb, out_b)
        assert np.array_equal(ga, ea) and np.array_equal(gb, eb), what
        for p in (out_a, out_b, d_w):
            ctx.free(p)
    assert seen["c2"][:2] == (128, 1536) and seen["c4"][:2] == (256, 768) and seen["canon2"][:2] == (128, 768), seen


def test_plain_c_resident_pipeline(tmp_path):
    """examples/resident_pipeline.c: a plain-C host with everything resident in HBM, its outputs first from plain device
    allocations, then from the pool, then FRESH from the pool for every launch; identical elements every way."""
    csrc = os.path.join(ROOT, "kmers.jl_amd", "csrc")
    exe = tmp_path / "resident_pipeline"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "resident_pipeline.c"),
                    "-L", csrc, "-lkmers_hip", f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    out = subprocess.run([str(exe), "300"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "plain allocations:" in out.stdout and "outputs from the pool:" in out.stdout and ": equal" in out.stdout, out.stdout
    assert "pool: kmers in 3 handles of 1 GiB" in out.stdout and "fresh outputs per launch:" in out.stdout, out.stdout
    assert "kmers only, plain block:" in out.stdout and "kmers only, by role:" in out.stdout
