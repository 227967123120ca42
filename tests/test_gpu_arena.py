"""The context's device-memory arena (include/kmers_hip.h: kmers_arena_reserve / _release / _info behind kmers_dev_alloc /
kmers_dev_free): sub-allocation, alignment, merging of freed ranges, fall-through to plain allocations, refusal to release
while blocks are out -- and that a launch whose source and outputs all live in the arena is bit-identical to the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# torch FIRST: it brings its own copy of the HIP / HSA runtime libraries, and a process that has already loaded /opt/rocm's
# (through libkmers_hip.so) cannot initialise torch's afterwards ("No HIP GPUs are available")
torch = pytest.importorskip("torch")


@pytest.fixture()
def km():
    import kmers_jl_amd
    return kmers_jl_amd


def need_map_memory(free_b, total_b):
    """The tests of the measured region map need 100 GB of free HBM.  On a device that HAS that much memory (MI355X: 288 GB) too
    little of it free is a failure of the test run (a leaked allocation, a second tenant), not a reason to skip: a skip reads as
    green.  Smaller devices skip."""
    if free_b >= 100e9:
        return
    if total_b >= 256e9:
        pytest.fail(f"a {total_b / 1e9:.0f} GB device with only {free_b / 1e9:.0f} GB free: the arena map tests need 100 GB")
    pytest.skip(f"needs 100 GB of free HBM for a map with more than one class, this device has {total_b / 1e9:.0f} GB")


def test_arena_suballocation_and_release(km):
    cap = km._capi
    ctx = km.Context(0)
    G = cap.ARENA_GRANULE
    assert ctx.arena_info() == (0, 0, 0)
    p_plain = ctx.alloc(1 << 20)                      # no arena yet: a plain allocation
    reserved = ctx.arena_reserve(64 * G + 5)          # rounded up to whole granules
    assert reserved == 65 * G and ctx.arena_info() == (65 * G, 0, 65 * G)
    assert ctx.lib.kmers_arena_reserve(ctx.handle, G) == cap.E_BADARG   # one arena per context
    a = ctx.alloc(3 * G - 7)
    b = ctx.alloc(10 * G)
    c = ctx.alloc(1)
    base = a
    assert (a - base, b - base, c - base) == (0, 3 * G, 13 * G) and a % G == 0
    assert ctx.arena_info() == (65 * G, 14 * G, 51 * G)
    assert ctx.lib.kmers_arena_release(ctx.handle) == cap.E_BADARG      # blocks are out
    assert "still allocated" in ctx.last_error()
    assert ctx.lib.kmers_dev_free(ctx.handle, C.c_void_p(b + 8)) == cap.E_BADARG   # not the start of a block
    ctx.free(b)
    assert ctx.arena_info() == (65 * G, 4 * G, 51 * G)
    d = ctx.alloc(9 * G)                               # best fit: the 10-granule hole, not the 51-granule tail
    assert d == b
    e = ctx.alloc(1 * G)                               # ... and the granule left of that hole
    assert e == b + 9 * G
    big = ctx.alloc(60 * G)                            # does not fit any more: falls through to a plain allocation
    assert not (base <= big < base + reserved)
    for p in (a, c, d, e, big, p_plain):
        ctx.free(p)
    assert ctx.arena_info() == (65 * G, 0, 65 * G)    # everything merged back into one range
    ctx.arena_release()
    assert ctx.arena_info() == (0, 0, 0)
    ctx.arena_release()                                # idempotent
    assert ctx.arena_reserve(2 * G) == 2 * G           # and a context can reserve again
    ctx.alloc(G)                                       # a block still out at destroy: the context frees the arena
    ctx.close()


def test_launch_inside_the_arena_matches_the_oracle(km, orc):
    cap = km._capi
    ctx = km.Context(0)
    ctx.arena_reserve(256 << 20)
    L, K, bits = 3_000_017, 31, 4
    nw = (L * bits + 63) // 64
    n = L - K + 1
    d_words, d_k, d_h = ctx.alloc(nw * 8 + 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
    r, u, f = ctx.arena_info()
    assert u >= (nw + 2 * n) * 8 and all(p % cap.ARENA_GRANULE == 0 for p in (d_words, d_k, d_h))
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 11, 0, nw, bits, 0, d_words), "kmers_synth_dna")
    seq = cap.Seq(d_words, L, 0, 0, bits, 0)
    res = cap.Result()
    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, d_k, d_h, 0, cap.MEM_DEVICE, C.byref(res))
    assert rc == 0 and res.n_out == n, ctx.last_error()
    kmers, hashes = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    ctx.d2h(kmers, d_k)
    ctx.d2h(hashes, d_h)
    ek, eh, eres = orc.canonical(orc.synth_words(11, 0, nw, bits), L, bits, 2, K)
    assert eres.status == 0 and np.array_equal(kmers, ek[:, 0]) and np.array_equal(hashes, eh)
    for p in (d_words, d_k, d_h):
        ctx.free(p)
    ctx.arena_release()
    ctx.close()


def test_arena_measures_its_region_map_and_spreads_the_outputs_of_a_launch(km):
    """A large arena measures the region map of its block (memory_api.hip: store streams inside one region class of HBM share a
    lower write rate than streams in different classes) and places the arrays of a launch accordingly: a block lies in another
    class than the block before it.  The map itself is a property of the machine: the test asks for its invariants only."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    reserved = ctx.arena_reserve(int(free_b * 0.7))
    base, gran, classes = ctx.arena_regions()
    assert base and gran == 4 << 30 and len(classes) == reserved // gran
    assert 2 <= len(set(classes)) <= 16, classes
    # classes come in runs (regions of tens of gigabytes), not as noise: most granules have a neighbour of their own class
    lonely = sum(1 for i in range(1, len(classes) - 1) if classes[i] != classes[i - 1] and classes[i] != classes[i + 1])
    assert lonely <= len(classes) // 8, classes

    def covered(ptr, nbytes):
        off = ptr - base
        assert 0 <= off and off + nbytes <= reserved
        return {classes[min(g, len(classes) - 1)] for g in range(off // gran, (off + nbytes - 1) // gran + 1)}
    small = ctx.alloc(64 << 20)          # (a small block first: the source words of a launch)
    a = ctx.alloc(8 << 30)
    b = ctx.alloc(8 << 30)
    assert ctx.arena_info()[1] == (64 << 20) + (16 << 30)
    assert all(p % cap.ARENA_GRANULE == 0 for p in (small, a, b))
    ca, cb = covered(a, 8 << 30), covered(b, 8 << 30)
    assert ca != cb or len(ca) >= 2, (ca, cb, classes)   # the outputs of one launch never share one single class
    # kmers_placement_probe (the same measurement, for buffers of the caller's): the pair the arena chose writes at the
    # two-class rate, the two halves of ONE of its blocks at the one-class rate
    apart, together = ctx.placement_probe(a, b, 1 << 30), ctx.placement_probe(a, a + (4 << 30), 1 << 30)
    assert 4000 < together < apart < 9000 and apart > 1.05 * together, (apart, together)
    res_bad = ctx.lib.kmers_placement_probe(ctx.handle, C.c_void_p(a + 8), C.c_void_p(b), 1 << 30, C.byref(C.c_double()))
    assert res_bad == cap.E_BADARG
    for p in (small, a, b):
        ctx.free(p)
    assert ctx.arena_info() == (reserved, 0, reserved)
    ctx.arena_release()
    # probing switched off: no map, best fit from the bottom of the block
    ctx.set_param(cap.PARAM_ARENA_NO_PROBE, 1)
    ctx.arena_reserve(reserved // 4)
    base2, gran2, classes2 = ctx.arena_regions()
    assert gran2 == 0 and classes2 == []
    p0, p1 = ctx.alloc(1 << 30), ctx.alloc(1 << 30)
    assert p0 == base2 and p1 == p0 + (1 << 30)
    ctx.close()


def test_lone_output_lies_across_a_class_boundary_and_is_written_through_two_windows(km, orc):
    """kmers_dev_alloc_role(KMERS_ALLOC_LONE_OUTPUT): the only output array of a launch is placed ACROSS a class boundary of the
    arena's map -- its two halves write at the two-class rate (kmers_placement_probe) -- and the launch that finds its one
    output there writes it in split order with its own launch shape: the results are those of the oracle, and of the same launch
    into an ordinary block, bit for bit (CanonicalKmers without hashes, SpacedKmers, a two-word kmer array, a tuple array)."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    reserved = ctx.arena_reserve(int(free_b * 0.7))
    base, gran, classes = ctx.arena_regions()
    lone = ctx.alloc(4 << 30, lone_output=True)
    plain = ctx.alloc(4 << 30)
    assert base <= lone < base + reserved and lone % cap.ARENA_GRANULE == 0
    halves, inside = ctx.placement_probe(lone, lone + (2 << 30), 1 << 30), ctx.placement_probe(plain, plain + (2 << 30), 1 << 30)
    assert halves > 1.05 * inside, (halves, inside, classes)
    ctx.free(lone)
    ctx.free(plain)
    res = cap.Result()
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    for bits, what in ((2, "canonical"), (4, "spaced"), (4, "fw4"), (4, "tuples")):
        L = 30_000_001 if what == "spaced" else 12_000_003     # (outputs of 64 MiB and more are placed by role)
        words = orc.synth_words(31 + bits, 0, (L * bits + 63) // 64 + 1, bits)
        d_w = ctx.alloc(words.nbytes)
        ctx.h2d(d_w, words)
        seq = cap.Seq(d_w, L, 0, 0, bits, 0)
        if what == "canonical":      # C3's shape
            exp = orc.canonical(words, L, bits, 2, 31)[0]
            call = lambda out: ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), 31, 2, out, None, 0, ASYNC, C.byref(res))
        elif what == "spaced":       # C5's
            exp = orc.spaced(words, L, bits, 2, 21, 3)[0]
            call = lambda out: ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), 21, 3, 2, out, ASYNC, C.byref(res))
        elif what == "fw4":          # two-word kmers of a 4-bit alphabet
            exp = orc.fw_kmers(words, L, bits, 4, 31)[0]
            call = lambda out: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 31, 4, out, None, ASYNC, C.byref(res))
        else:                        # Tuple{Kmer,Kmer} elements of FwRvIterator{63}
            f, r, _ = orc.fwrv(words, L, bits, 2, 63)
            exp = np.concatenate([f, r], axis=1)
            call = lambda out: ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 63, 2, out, None, ASYNC | cap.OUT_TUPLES, C.byref(res))
        nbytes = exp.nbytes
        assert nbytes >= 64 << 20
        got = {}
        for role in (True, False):
            out = ctx.alloc(nbytes, lone_output=role)
            assert call(out) == 0, ctx.last_error()
            assert ctx.sync()[0] == 0
            host = np.zeros(exp.shape, np.uint64)
            ctx.d2h(host, out)
            got[role] = host
            ctx.free(out)
        assert np.array_equal(got[True], exp) and np.array_equal(got[False], exp), what
        ctx.free(d_w)
    ctx.close()


def test_lone_output_launches_fuzzed(km, orc):
    """Random geometries through the lone-output path (array across a class boundary, split order, its own launch shapes):
    kmer widths of one to four words, both kmer alphabets, strides, views that start anywhere in a word, odd lengths -- every
    element against the oracle."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    ctx.arena_reserve(int(free_b * 0.7))
    rng = np.random.default_rng(2024)
    res = cap.Result()
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    for case in range(14):
        bits = int(rng.choice([2, 4]))
        dst = int(rng.choice([2, 2, 4]))
        K = int(rng.choice([1, 5, 21, 31, 32, 33, 63, 64])) if dst == 2 else int(rng.choice([3, 16, 17, 31, 32]))
        what = str(rng.choice(["canonical", "fw", "spaced", "tuples"]))
        J = int(rng.choice([2, 3, 5, 16])) if what == "spaced" else 1
        N = (K * dst + 63) // 64
        per = 2 * N if what == "tuples" else N
        n_min = (64 << 20) // (8 * per) + 1000                     # outputs of 64 MiB and more are placed by role
        L = (n_min + int(rng.integers(0, 300_000))) * J + K
        first = int(rng.choice([0, 1, 7, 15, 16, 31, 33]))
        words = orc.synth_words(case, 0, ((first + L) * bits + 63) // 64 + 1, bits)
        d_w = ctx.alloc(words.nbytes)
        ctx.h2d(d_w, words)
        seq = cap.Seq(d_w, L, first, 0, bits, 0)
        # the oracle takes a view that starts at symbol 0: shift the words
        shifted = np.zeros(len(words), np.uint64)
        sh = first * bits
        w0, b0 = sh // 64, sh % 64
        src64 = words[w0:]
        shifted[:len(src64)] = src64 >> np.uint64(b0)
        if b0:
            shifted[:len(src64) - 1] |= src64[1:] << np.uint64(64 - b0)
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
        out = ctx.alloc(exp.nbytes, lone_output=True)
        assert call(out) == 0, ctx.last_error()
        assert ctx.sync()[0] == 0
        host = np.zeros(exp.shape, np.uint64)
        ctx.d2h(host, out)
        ctx.free(out)
        ctx.free(d_w)
        assert np.array_equal(host, exp), (case, bits, dst, K, what, J, L, first)
    ctx.close()


def test_two_output_launches_placed_by_the_arena_and_shaped_by_the_launcher(km, orc):
    """The launch shapes of the headline path (csrc/stream_launch.hpp: 128 x 1536 for two one-word arrays, 256 x 768 for two-word
    kmers + reverse complements, 128 x 768 for two-word canonical kmers + hashes) are picked by the LAUNCHER when it sees its
    two output arrays well placed in the arena's map.  Here nothing is forced: a mapped arena, kmers_dev_alloc for every
    buffer, the library's own choice (kmers_last_launch_shape says what it was) -- and EVERY element of 64 MiB and more per
    array against the oracle: C2 (CanonicalDNAMers{31} + fx_hash), C4 (FwDNAMers{63} + reverse complements), two-word
    canonical kmers + hashes (src/iterators/CanonicalKmers.jl:131-144, :220-225; src/kmer.jl:255-261)."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    ctx.arena_reserve(int(free_b * 0.7))
    best, one = ctx.arena_rates()
    assert best > 1.05 * one > 0, (best, one)
    res = cap.Result()
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    seen = {}
    for what, K, L in (("c2", 31, 9_000_031), ("c4", 63, 4_500_063), ("canon2", 63, 4_500_063)):
        words = orc.synth_words(41 + K, 0, (L * 4 + 63) // 64 + 1, 4)
        n = L - K + 1
        N = (2 * K + 63) // 64
        assert 8 * n * N >= 64 << 20
        out_a = ctx.alloc(8 * n * N)       # the two outputs one after the other: the arena puts them into two classes
        out_b = ctx.alloc(8 * n * (N if what == "c4" else 1))
        d_w = ctx.alloc(words.nbytes)
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
    # the table's shapes for well-placed arrays (a map in which the arena could NOT place two arrays well would leave the base
    # rule's 256 threads: then the placement, not the launcher, is what this run could not exercise -- say so loudly)
    # (KMERS_PARAM_SHAPE_CALIBRATE, on by default, lets the first large launch time the table's shape against the base rule's and
    # keep the faster one: on most boxes the table's)
    assert seen["c2"][:2] in ((128, 1536), (256, 1024)) and seen["c4"][:2] in ((256, 768), (256, 512)) and seen["canon2"][:2] in ((128, 768), (256, 512), (256, 256)), seen
    ctx.close()


def test_c2_one_gbase_in_the_arena_all_element_identities(km):
    """C2 at its full size with the buffers from a mapped arena and the launcher's own shape: the three identities that tie every
    one of the 999 999 970 elements to its neighbours and to the hash definition (the oracle pins the head and the tail in
    tests/test_gpu_configs.py; here the point is the arena's placement + 128 x 1536 at full size)."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    ctx.arena_reserve(int(free_b * 0.7))
    dev = torch.device("cuda", 0)
    L, K = 1_000_000_000, 31
    n = L - K + 1
    nw = (L * 4 + 63) // 64
    p_k, p_h, p_w = ctx.alloc(8 * n), ctx.alloc(8 * n), ctx.alloc(8 * nw + 16)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 0x9E3779B97F4A7C15 ^ 2, 0, nw, 4, 0, p_w), "kmers_synth_dna")
    seq = cap.Seq(p_w, L, 0, 0, 4, 0)
    res = cap.Result()
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, p_k, p_h, 0, cap.MEM_DEVICE, C.byref(res)) == 0, ctx.last_error()
    assert res.n_out == n and ctx.last_launch_shape()[:2] in ((128, 1536), (256, 1024)), ctx.last_launch_shape()

    class Raw:
        def __init__(self, ptr, words):
            self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None}
    km_t, h_t = torch.as_tensor(Raw(p_k, n), device=dev), torch.as_tensor(Raw(p_h, n), device=dev)
    FX = torch.tensor(0x517CC1B727220A95, dtype=torch.int64, device=dev)
    CH = 1 << 27
    fold = 0
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        assert bool(torch.equal(h_t[lo:hi], km_t[lo:hi] * FX)), lo              # fx_hash(x) = x * FX_CONSTANT for one-word kmers
        assert int((km_t[lo:hi] >> 62).ne(0).sum().item()) == 0, lo             # unused top bits zero (src/kmer.jl:32-44)
        t = km_t[lo:hi]
        while t.numel() > 1:
            h2 = t.numel() // 2
            rest = t[2 * h2:]
            t = torch.bitwise_xor(t[:h2], t[h2:2 * h2])
            if rest.numel():
                t = torch.cat([t, rest])
        fold ^= int(t.item()) & (2**64 - 1)
    xr = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)) == 0
    assert fold == xr.value                                                     # every element present exactly once (another kernel's fold)
    for p in (p_k, p_h, p_w):
        ctx.free(p)
    ctx.close()


def test_contexts_of_one_device_share_one_arena(km):
    """One arena per device and process: a second context that calls kmers_arena_reserve ATTACHES to the block the first one
    reserved (it does not get a second three quarters of the device), both sub-allocate from it (thread-safely: one context per
    host thread is the model, tests/test_gpu_threads.py), and the block lives until the last context has let go."""
    import threading
    cap = km._capi
    G = cap.ARENA_GRANULE
    a, b = km.Context(0), km.Context(0)
    assert a.arena_reserve(64 * G) == 64 * G
    assert b.arena_reserve(0) == 64 * G                       # attached: not a reservation of its own
    assert b.lib.kmers_arena_reserve(b.handle, G) == cap.E_BADARG   # ... and attached only once
    pa, pb = a.alloc(3 * G), b.alloc(5 * G)
    assert pb == pa + 3 * G and a.arena_info() == b.arena_info() == (64 * G, 8 * G, 56 * G)
    got = []

    def worker(ctx):
        mine = []
        for _ in range(200):
            mine.append(ctx.alloc(G))
            if len(mine) > 8:
                ctx.free(mine.pop(0))
        got.append(mine)
    ts = [threading.Thread(target=worker, args=(c,)) for c in (a, b)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    live = [p for mine in got for p in mine]
    assert len(set(live)) == len(live) == 16                   # no block handed out twice
    for mine, ctx in zip(got, (a, b)):
        for p in mine:
            ctx.free(p)
    a.free(pa)
    assert a.lib.kmers_arena_release(a.handle) == 0            # a lets go: the block stays for b (whose 5 granules are still out)
    assert a.arena_info() == (0, 0, 0) and b.arena_info() == (64 * G, 5 * G, 56 * G)   # (b's block lies between the two free ranges)
    assert b.lib.kmers_arena_release(b.handle) == cap.E_BADARG   # the LAST one cannot release while blocks are out
    b.free(pb)
    b.arena_release()
    assert b.arena_info() == (0, 0, 0)
    assert a.arena_reserve(2 * G) == 2 * G                     # and the device can get a new arena afterwards
    a.close()
    b.close()


def test_plain_c_resident_pipeline(km, orc, tmp_path):
    """examples/resident_pipeline.c: a plain-C host with everything resident in HBM, its outputs first from plain device
    allocations, then from the context's arena; identical elements either way, checked here against the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "kmers.jl_amd", "csrc")
    exe = tmp_path / "resident_pipeline"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "resident_pipeline.c"),
                    "-L", csrc, "-lkmers_hip", f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    out = subprocess.run([str(exe), "64"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "plain allocations:" in out.stdout and "outputs from the arena:" in out.stdout and ": equal" in out.stdout, out.stdout
    assert "arena:" in out.stdout and "classes A" in out.stdout
    assert "kmers only, plain block:" in out.stdout and "kmers only, by role:" in out.stdout


def test_launch_shape_calibration_keeps_results_and_remembers(km):
    """KMERS_PARAM_SHAPE_CALIBRATE (stream_launch.hpp): the first launch of 1 GB or more into a pair of well-placed arena arrays
    times the table's shape against the base rule's and keeps the faster one for those arrays.  The outputs are the same
    whichever shape wrote them (XOR folds of both arrays against a run with the calibration off), the choice is one of the two
    candidates and stays, and with the calibration off the launcher uses its table."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    dev = torch.device("cuda", 0)
    ctx = km.Context(0)
    ctx.arena_reserve(int(free_b * 0.7))
    K, L = 31, 140_000_030
    n = L - K + 1
    nw = (L * 4 + 63) // 64
    p_k, p_h, p_w = ctx.alloc(8 * n), ctx.alloc(8 * n), ctx.alloc(8 * (nw + 2))
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 99, 0, nw, 4, 0, p_w), "kmers_synth_dna")
    seq = cap.Seq(p_w, L, 0, 0, 4, 0)
    res = cap.Result()

    class Raw:
        def __init__(self, ptr, words):
            self.__cuda_array_interface__ = {"shape": (words,), "typestr": "<i8", "data": (ptr, False), "version": 2, "strides": None}
    km_t, h_t = torch.as_tensor(Raw(p_k, n), device=dev), torch.as_tensor(Raw(p_h, n), device=dev)

    def fold(t):
        acc = 0
        for i in range(0, n, 1 << 26):
            c = t[i:i + (1 << 26)]
            while c.numel() > 1:
                h = c.numel() // 2
                c = torch.cat((c[:h] ^ c[h:2 * h], c[2 * h:]))
            acc ^= int(c.item()) & 0xFFFFFFFFFFFFFFFF
        return acc

    def run():
        km_t.zero_()
        h_t.zero_()
        torch.cuda.synchronize()
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, p_k, p_h, 3, cap.MEM_DEVICE, C.byref(res)) == 0, ctx.last_error()
        assert res.n_out == n
        return ctx.last_launch_shape()[:2], fold(km_t), fold(h_t)

    ctx.set_param(cap.PARAM_SHAPE_CALIBRATE, 0)
    table_shape, fk, fh = run()
    assert table_shape in ((128, 1536), (256, 1024)), table_shape   # (256 x 1024: the arena could not place the pair well on this box)
    ctx.set_param(cap.PARAM_SHAPE_CALIBRATE, 1)
    first = run()
    assert first[0] in (table_shape, (256, 1024)) and first[1:] == (fk, fh), (first, table_shape)
    t_ms, r_ms, rule = ctx.last_shape_calibration()
    if table_shape != (256, 1024):                                 # (the table departs from the rule: both were timed)
        assert 0 < t_ms < 50 and 0 < r_ms < 50 and rule == (first[0] == (256, 1024) != table_shape) and (not rule or r_ms < 0.97 * t_ms), (t_ms, r_ms, rule)
    again = run()
    assert again == first                                          # remembered: the same shape, the same outputs
    ctx.set_param(cap.PARAM_SHAPE_CALIBRATE, 0)
    assert run() == (table_shape, fk, fh)
    assert ctx.last_shape_calibration() == (0.0, 0.0, False)
    ctx.close()


def test_alloc_launch_free_loop_calibrates_at_most_once_and_never_inside_an_async_call(km):
    """the reference's collect allocates fresh outputs per call (src/iterators/CanonicalKmers.jl:199-225 +
    Base.collect).  The launcher's remembered shape is keyed by the launch configuration and the PLACEMENT of the arrays (the
    runs of the arena they start in), survives kmers_dev_free, and a KMERS_ASYNC call never blocks to measure: a loop of
    {alloc, launch, free} over the arena calibrates once, not once per call."""
    cap = km._capi
    free_b, total_b = torch.cuda.mem_get_info(0)
    need_map_memory(free_b, total_b)
    ctx = km.Context(0)
    ctx.arena_reserve(int(free_b * 0.7))
    K, L = 31, 140_000_030
    n = L - K + 1
    nw = (L * 4 + 63) // 64
    p_w = ctx.alloc(8 * (nw + 2))
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 7, 0, nw, 4, 0, p_w), "kmers_synth_dna")
    seq = cap.Seq(p_w, L, 0, 0, 4, 0)
    res = cap.Result()
    # asynchronous launches first: whatever the table says, nothing is timed
    p_k, p_h = ctx.alloc(8 * n), ctx.alloc(8 * n)
    for _ in range(3):
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, p_k, p_h, 0, cap.MEM_DEVICE | cap.ASYNC, C.byref(res)) == 0, ctx.last_error()
    assert ctx.sync()[0] == 0 and ctx.shape_calibrations() == 0
    ctx.free(p_k)
    ctx.free(p_h)
    shapes = set()
    for it in range(6):
        p_k, p_h = ctx.alloc(8 * n), ctx.alloc(8 * n)   # fresh blocks every time (the same places: the arena is deterministic)
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, p_k, p_h, 0, cap.MEM_DEVICE, C.byref(res)) == 0, ctx.last_error()
        shapes.add(ctx.last_launch_shape()[:2])
        ctx.free(p_k)
        ctx.free(p_h)
    assert ctx.shape_calibrations() <= 1, ctx.shape_calibrations()
    assert len(shapes) == 1, shapes   # ... and what was measured once is what every later call runs
    ctx.close()
