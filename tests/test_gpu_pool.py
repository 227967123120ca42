"""The device's class pool (csrc/pool_api.hip) on hardware: the virtual-memory calls it rests on behave as the code assumes,
blocks are assembled by region class (the arrays of a launch differ at every position, a lone output's halves differ), data
survives, memory is reused and returned, and launches into blocks of the pool give the oracle's results without any shape
calibration (reference: src/iterators/CanonicalKmers.jl:199-225, src/kmer.jl:255-261)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# torch FIRST (tests/test_gpu_arena.py says why)
torch = pytest.importorskip("torch")

import kmers_jl_amd as km  # noqa: E402
from kmers_jl_amd import _capi as cap  # noqa: E402

MiB, GiB = 1 << 20, 1 << 30


@pytest.fixture()
def ctx():
    c = km.Context(0)
    yield c
    c.close()


def h2d(ctx, ptr, arr):
    ctx.check(ctx.lib.kmers_memcpy_h2d(ctx.handle, C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes), "h2d")


def d2h(ctx, ptr, n):
    back = np.zeros(n, dtype=np.uint64)
    ctx.check(ctx.lib.kmers_memcpy_d2h(ctx.handle, back.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), back.nbytes), "d2h")
    return back


def test_vmm_calls_behave_as_the_pool_assumes(ctx):
    # KMERS_OK = fresh mappings show their memory and a re-used range shows its NEW memory after the pool's flush; the flag says
    # whether the stale-translation behaviour the flush exists for was reproduced (it is on ROCm 7.2; either answer is fine)
    assert ctx.pool_selftest() in (True, False)
    ctx.pool_trim()


def test_blocks_are_assembled_by_class_and_hold_their_data(ctx):
    na, nb = 6 * GiB + 5 * MiB, 6 * GiB
    a = ctx.alloc(na)
    b = ctx.alloc(nb)
    chunk, ca = ctx.pool_layout(a)
    _, cb = ctx.pool_layout(b)
    assert chunk == GiB and len(ca) == 7 and len(cb) == 6
    info = ctx.pool_info()
    assert info["in_use"] == 13 * GiB and info["held"] >= info["in_use"] + info["n_classes"] * GiB
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
    assert len(co) == 10 and len(set(co[:5])) == 1 and all(co[i] != co[i + 5] for i in range(5)), co
    half = -(-(odd_bytes // 2) // 4096) * 4096                 # the second half begins on the boundary between handle 4 and handle 5
    assert odd % 4096 == 0 and ctx.placement_probe(odd, odd + half, GiB) > 6700 and ctx.placement_probe(odd + half - GiB, odd + half, GiB) > 6700
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(odd + 4096)) == cap.E_BADARG
    ctx.free(odd)
    # a pattern across every chunk boundary, written and read through the C ABI's copies
    rng = np.random.default_rng(11)
    for off in (0, GiB - 4096, 3 * GiB - 8, na - 8192):
        data = rng.integers(0, 1 << 63, 1024, dtype=np.uint64)
        h2d(ctx, a + off, data)
        assert np.array_equal(data, d2h(ctx, a + off, 1024))
    for p in (a, b, lone):
        ctx.free(p)
    assert ctx.pool_info()["in_use"] == 0


def test_alloc_free_loop_reuses_memory_and_never_shows_stale_data(ctx):
    # the reference's collect: one array per call.  Sizes change from call to call so that chunks move between address ranges;
    # what a block shows must always be what was last written through IT.
    rng = np.random.default_rng(5)
    held = []
    for it in range(10):
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
        held.append(ctx.pool_info()["held"])
    assert ctx.pool_info()["in_use"] == 0
    assert held[-1] <= 12 * GiB + 4 * GiB + 64 * GiB, held   # never more than the largest round + yardsticks + the search budget
    assert held[-1] == held[-3], "the pool kept growing although every block of every round had been returned"
    released = ctx.pool_trim()
    assert released == held[-1] and ctx.pool_info()["held"] == 0


def test_small_blocks_and_switched_off_pool_are_plain_allocations(ctx):
    p = ctx.alloc(cap.POOL_MIN_BYTES - 1)
    assert ctx.pool_layout(p)[1] == []
    ctx.free(p)
    ctx.set_param(cap.PARAM_POOL, 0)
    q = ctx.alloc(2 * GiB)
    assert ctx.pool_layout(q)[1] == [] and ctx.pool_info()["in_use"] == 0
    ctx.free(q)
    ctx.set_param(cap.PARAM_POOL, 1)
    r = ctx.alloc(GiB)
    assert len(ctx.pool_layout(r)[1]) == 1
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(r + 4096)) == cap.E_BADARG   # not the start of the block
    ctx.free(r)


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


def test_launches_into_pool_blocks_match_the_oracle_and_never_calibrate(ctx, orc):
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
    assert np.array_equal(hashes, kmers * np.uint64(0x517cc1b727220a95))   # every element: fx_hash of one word, seed 0 (src/kmer.jl:255-261)
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
    assert ctx.shape_calibrations() == 0
    for p in (d_words, d_k, d_h, blk):
        ctx.free(p)
