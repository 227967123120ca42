"""The context's device-memory arena (include/kmers_hip.h: kmers_arena_reserve / _release / _info behind kmers_dev_alloc /
kmers_dev_free): sub-allocation, alignment, merging of freed ranges, fall-through to plain allocations, refusal to release
while blocks are out -- and that a launch whose source and outputs all live in the arena is bit-identical to the oracle."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def km():
    import kmers_jl_amd
    return kmers_jl_amd


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
