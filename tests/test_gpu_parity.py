"""GPU parity tests proper: HIP kernels through the C ABI (libkmers_hip.so) vs the CPU oracle on
the same seeded inputs, bit-exact (all arithmetic is unsigned 64-bit integer).  Run with -m gpu."""
import ctypes as C

import numpy as np
import pytest

import naive

pytestmark = pytest.mark.gpu

KS = [1, 2, 5, 16, 21, 31, 32, 33, 47, 63, 64, 65, 96, 128, 129, 200]  # 129, 200: more than four words (run-time-width kernel)


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    return kmers_jl_amd


@pytest.fixture(scope="module")
def ctx(km):
    c = km.Context(0)
    yield c
    c.close()


def vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def make_seq(km, words, n_bases, bits, first_base=0, origin=0):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    return km._capi.Seq(words.ctypes.data, n_bases, first_base, origin, bits, 0), words


def run_canonical(km, ctx, words, L, bits, K, seed=0, first_base=0, device=False):
    cap = km._capi
    N = (2 * K + 63) // 64
    n = max(0, L - K + 1)
    kmers = np.zeros((max(n, 1), N), dtype=np.uint64)
    hashes = np.zeros(max(n, 1), dtype=np.uint64)
    res = cap.Result()
    if not device:
        seq, keep = make_seq(km, words, L, bits, first_base)
        rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, vp(kmers), vp(hashes), seed, cap.MEM_HOST,
                                     C.byref(res))
    else:
        words = np.ascontiguousarray(words, dtype=np.uint64)
        dw = ctx.alloc(words.nbytes + 8)
        dk, dh = ctx.alloc(kmers.nbytes), ctx.alloc(hashes.nbytes)
        ctx.h2d(dw, words)
        seq = cap.Seq(dw, L, first_base, 0, bits, 0)
        rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, dk, dh, seed, cap.MEM_DEVICE, C.byref(res))
        if rc == 0 and n:
            ctx.d2h(kmers, dk)
            ctx.d2h(hashes, dh)
        for p in (dw, dk, dh):
            ctx.free(p)
    return rc, kmers[:n], hashes[:n], res


@pytest.mark.parametrize("bits", [2, 4])
@pytest.mark.parametrize("device", [False, True])
def test_canonical_hash_parity(km, ctx, orc, bits, device):
    """CanonicalDNAMers{K} + fx_hash vs oracle (CanonicalKmers.jl:131-144,:220-225; kmer.jl:255-261)."""
    for K in KS:
        for L in (0, K - 1, K, K + 1, 97, 4096 + K - 1, 4096 + K, 70001):
            if L < 0:
                continue
            seed = (K * 1315423911 + L) & (2**64 - 1)
            nwords = (L * bits + 63) // 64 + 1
            words = orc.synth_words(1234 + K, 3, nwords, bits)
            rc, kmers, hashes, res = run_canonical(km, ctx, words, L, bits, K, seed, device=device)
            assert rc == 0, (K, L, ctx.last_error())
            ek, eh, eres = orc.canonical(words, L, bits, 2, K, seed=seed)
            assert res.n_out == eres.n_out == len(ek)
            assert np.array_equal(kmers, ek), (K, L)
            assert np.array_equal(hashes, eh), (K, L)


@pytest.mark.parametrize("bits", [2, 4])
def test_fw_and_revcomp_parity(km, ctx, orc, bits):
    """FwKmers / FwRvIterator (FwKmers.jl:88-115, CanonicalKmers.jl:94-144) incl. two-word kmers."""
    cap = km._capi
    for K in KS:
        N = (2 * K + 63) // 64
        for L in (K, K + 7, 5000, 40000):
            words = orc.synth_words(99 + K, 0, (L * bits + 63) // 64 + 1, bits)
            n = L - K + 1
            fw = np.zeros((n, N), dtype=np.uint64)
            rv = np.zeros((n, N), dtype=np.uint64)
            res = cap.Result()
            seq, keep = make_seq(km, words, L, bits)
            rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res))
            assert rc == 0
            efw, erv, _ = orc.fwrv(words, L, bits, 2, K)
            assert np.array_equal(fw, efw), (K, L)
            assert np.array_equal(rv, erv), (K, L)
            fw2 = np.zeros((n, N), dtype=np.uint64)
            rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, vp(fw2), None, cap.MEM_HOST, C.byref(res))
            assert rc == 0 and np.array_equal(fw2, efw)


@pytest.mark.parametrize("src,dst", [(2, 2), (4, 2), (2, 4), (4, 4)])
def test_all_recoding_schemes_and_wide_kmers(km, ctx, orc, src, dst):
    """Every RecodingScheme between 2- and 4-bit alphabets (construction.jl:75-100: Copyable,
    FourToTwo, TwoToFour) and kmers of 1..4 words, as in test/runtests.jl:674-690, :749-760."""
    cap = km._capi
    per = 64 // dst
    ks = sorted({1, 3, per - 1, per, per + 1, 2 * per - 1, 2 * per, 2 * per + 1, 3 * per, 3 * per + 1, 4 * per - 1, 4 * per})
    rng = np.random.default_rng(src * 10 + dst)
    for K in ks:
        N = (K * dst + 63) // 64
        for L in (K, K + 9, 3000, 17001):
            if src == 4 and dst == 4:  # Copyable 4 -> 4 keeps ambiguous symbols
                text = naive.random_text(rng, L, p_amb=0.1)
                words = naive.longseq_words(text, 4)
            else:
                words = orc.synth_words(K + L, 2, (L * src + 63) // 64 + 1, src)
            n = L - K + 1
            fw = np.zeros((n, N), dtype=np.uint64)
            rv = np.zeros((n, N), dtype=np.uint64)
            ck = np.zeros((n, N), dtype=np.uint64)
            hs = np.zeros(n, dtype=np.uint64)
            res = cap.Result()
            seq, keep = make_seq(km, words, L, src)
            rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res))
            assert rc == 0, (K, L, ctx.last_error())
            rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(ck), vp(hs), 17, cap.MEM_HOST, C.byref(res))
            assert rc == 0, (K, L, ctx.last_error())
            efw, erv, _ = orc.fwrv(words, L, src, dst, K)
            ek, eh, _ = orc.canonical(words, L, src, dst, K, seed=17)
            assert np.array_equal(fw, efw), (K, L)
            assert np.array_equal(rv, erv), (K, L)
            assert np.array_equal(ck, ek) and np.array_equal(hs, eh), (K, L)
        # spaced, tile and gather paths
        for J in (3, 40):
            L = 5000
            words = orc.synth_words(K, 0, (L * src + 63) // 64 + 1, src)
            n = (L - K) // J + 1
            out = np.zeros((n, N), dtype=np.uint64)
            res = cap.Result()
            seq, keep = make_seq(km, words, L, src)
            rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(out), cap.MEM_HOST, C.byref(res))
            assert rc == 0, (K, J, ctx.last_error())
            ek, _ = orc.spaced(words, L, src, dst, K, J)
            assert np.array_equal(out, ek), (K, J)


def test_kmers_wider_than_four_words(km, ctx, orc):
    """Kmer{A,K,N} has no upper bound on N (src/kmer.jl:97-111): K > 128 (2-bit) / K > 64 (4-bit) run on the run-time-width
    kernel.  FwKmers + reverse complements, CanonicalKmers + fx_hash and SpacedKmers against the oracle (N <= 8), every
    recoding scheme incl. text, and the EncodeError position; beyond the oracle's widths against the independent naive
    slicer (big-integer packing from the layout rule only)."""
    cap = km._capi
    rng = np.random.default_rng(4242)
    for src, dst, ks in ((2, 2, (129, 160, 256)), (4, 2, (129, 193, 255)), (4, 4, (65, 100, 128)), (2, 4, (65, 127)), (8, 2, (130,)), (8, 4, (70,))):
        for K in ks:
            N = (K * dst + 63) // 64
            assert N > 4
            for L in (K, K + 3, 1500):
                text = naive.random_text(rng, L, p_amb=0.05 if (src == 4 and dst == 4) or (src == 8 and dst == 4) else 0.0)
                words = naive.ascii_words(text) if src == 8 else naive.longseq_words(text, src)
                osrc = 8 if src == 8 else src
                seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                n = L - K + 1
                res = cap.Result()
                fw, rv = np.zeros((n, N), np.uint64), np.zeros((n, N), np.uint64)
                assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res)) == 0
                efw, erv, _ = orc.fwrv(words, L, osrc, dst, K)
                assert np.array_equal(fw, efw) and np.array_equal(rv, erv), (src, dst, K, L)
                ck, hs = np.zeros((n, N), np.uint64), np.zeros(n, np.uint64)
                assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(ck), vp(hs), 77, cap.MEM_HOST, C.byref(res)) == 0
                ek, eh, _ = orc.canonical(words, L, osrc, dst, K, seed=77)
                assert np.array_equal(ck, ek) and np.array_equal(hs, eh), (src, dst, K, L)
                for J in (7, K + 5):
                    m = (L - K) // J + 1
                    out = np.zeros((m, N), np.uint64)
                    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(out), cap.MEM_HOST, C.byref(res)) == 0
                    es, _ = orc.spaced(words, L, osrc, dst, K, J)
                    assert np.array_equal(out, es), (src, dst, K, L, J)
            if dst == 2 and src in (4, 8):   # the first symbol the 2-bit alphabet cannot hold, in sequence order
                L = 1200
                t = list(naive.random_text(rng, L))
                t[700], t[333], t[1100] = "N", "W", "-"
                text = "".join(t)
                words = naive.ascii_words(text) if src == 8 else naive.longseq_words(text, src)
                seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                out = np.zeros((L - K + 1, N), np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(out), None, 0, cap.MEM_HOST, C.byref(res))
                _, _, eres = orc.canonical(words, L, 8 if src == 8 else src, dst, K)
                assert (rc, res.err_pos, res.err_enc) == (1, eres.err_pos, eres.err_enc) == (1, 334, res.err_enc)
    # wider than the oracle goes: K = 700 two-bit (22 words), against the naive slicer
    K, L = 700, 1000
    text = naive.random_text(rng, L)
    words = naive.longseq_words(text, 4)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    N = (2 * K + 63) // 64
    fw, rv = np.zeros((L - K + 1, N), np.uint64), np.zeros((L - K + 1, N), np.uint64)
    res = cap.Result()
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res)) == 0
    assert [tuple(int(x) for x in r) for r in fw] == naive.fw_kmers(text, K, 2)
    assert [tuple(int(x) for x in r) for r in rv] == [b for _, b in naive.fwrv(text, K, 2)]
    # (the other entry points take these widths too: tests/test_gpu_wide.py)


def ascii_seq(km, text_or_bytes, L, alphabet=0, first_base=0):
    words = naive.ascii_words(text_or_bytes)
    return km._capi.Seq(words.ctypes.data, L, first_base, 0, 8, alphabet), words


def test_ascii_sources(km, ctx, orc):
    """AsciiEncode sources (String / Vector{UInt8}): FwKmers.jl:117-129, CanonicalKmers.jl:146-174,
    SpacedKmers.jl:109-139, UnambiguousKmers.jl:109-132; same results and the same EncodeError
    (position, offending byte) as the oracle."""
    cap = km._capi
    rng = np.random.default_rng(88)
    for rna in (0, 1):
        for dst in (2, 4):
            for K in (1, 4, 16, 17, 31, 33, 64):
                N = (K * dst + 63) // 64
                for L in (K - 1, K, K + 11, 4099, 20011):
                    if L < 0:
                        continue
                    text = naive.random_text(rng, L, p_amb=0.05 if dst == 4 else 0.0)
                    text = "".join(c.lower() if rng.random() < 0.3 else c for c in text)
                    if rna:
                        text = text.replace("T", "U").replace("t", "u")
                    seq, keep = ascii_seq(km, text, L, rna)
                    n = max(0, L - K + 1)
                    fw = np.zeros((max(n, 1), N), np.uint64)
                    rv = np.zeros((max(n, 1), N), np.uint64)
                    ck = np.zeros((max(n, 1), N), np.uint64)
                    hs = np.zeros(max(n, 1), np.uint64)
                    res = cap.Result()
                    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(fw), vp(rv), 0, C.byref(res)) == 0
                    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(ck), vp(hs), 5, 0, C.byref(res)) == 0
                    efw, erv, er = orc.fwrv(keep, L, 8 + rna, dst, K)
                    ek, eh, _ = orc.canonical(keep, L, 8 + rna, dst, K, seed=5)
                    assert er.status == 0
                    assert np.array_equal(fw[:n], efw) and np.array_equal(rv[:n], erv), (rna, dst, K, L)
                    assert np.array_equal(ck[:n], ek) and np.array_equal(hs[:n], eh)
    # spaced: tile path and gather path
    for K, J in ((3, 2), (21, 3), (5, 40), (33, 33)):
        L = 9000
        text = naive.random_text(rng, L)
        seq, keep = ascii_seq(km, text, L)
        n = (L - K) // J + 1
        N = (2 * K + 63) // 64
        out = np.zeros((n, N), np.uint64)
        res = cap.Result()
        assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, vp(out), 0, C.byref(res)) == 0
        ek, _ = orc.spaced(keep, L, 8, 2, K, J)
        assert np.array_equal(out, ek)
    # offset view into a byte buffer (unaligned host pointer arithmetic is the library's job)
    text = naive.random_text(rng, 5000)
    for first in (1, 3, 8, 13, 1001):
        seq, keep = ascii_seq(km, text, 3000, 0, first)
        out = np.zeros((3000 - 30, 1), np.uint64)
        res = cap.Result()
        assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 31, 2, vp(out), None, 0, C.byref(res)) == 0
        ek, _ = orc.fw_kmers(naive.ascii_words(text[first:first + 3000]), 3000, 8, 2, 31)
        assert np.array_equal(out, ek), first


def test_host_pointer_calls_in_chunks(km, ctx, orc):
    """Host-pointer calls with outputs of 96 MiB or more run in chunks (iterators_api.hip, run_chunked: the kernel of chunk c + 1
    beside the copy of chunk c): every element of a call that spans four chunks equals the oracle's and the one-launch path's
    (KMERS_PARAM_HOST_CHUNKS = -1), for CanonicalDNAMers{31} + fx_hash, FwDNAMers{63} + reverse complements (two-word elements) and
    SpacedDNAMers{21,3}; an ambiguous symbol in the third chunk gives the reference's EncodeError position
    (src/iterators/CanonicalKmers.jl:131-144, :220-225; SpacedKmers.jl:121-139)."""
    cap = km._capi
    res = cap.Result()
    # canonical + hashes: 13.5 M elements -> 2 x 108 MB -> four chunks of 4 Mi elements
    L, K = 13_500_000, 31
    words = orc.synth_words(77, 0, L // 16 + 2, 4)
    seq, keep = make_seq(km, words, L, 4)
    n = L - K + 1
    ek, eh, _ = orc.canonical(words, L, 4, 2, K, seed=9)
    for chunks in (0, -1):
        ctx.set_param(cap.PARAM_HOST_CHUNKS, chunks)
        k_out, h_out = np.zeros((n, 1), np.uint64), np.zeros(n, np.uint64)
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, vp(k_out), vp(h_out), 9, cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
        assert res.n_out == n and np.array_equal(k_out, ek) and np.array_equal(h_out, eh), chunks
    ctx.set_param(cap.PARAM_HOST_CHUNKS, 0)
    # an N in the third chunk (and a later one): the first one is reported, 1-based
    bad = words.copy()
    for pos in (9_000_123, 12_000_000):
        bad[(pos * 4) >> 6] |= np.uint64(0xF) << np.uint64((pos * 4) & 63)
    seqb, keepb = make_seq(km, bad, L, 4)
    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seqb), K, 2, vp(k_out), vp(h_out), 9, cap.MEM_HOST, C.byref(res))
    assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, 9_000_124, 0xF)
    # two-word elements, two arrays: 6.5 M x 16 B -> 2 x 104 MB -> four chunks of 2 Mi elements; a view that starts inside a word
    # (first_base = 5: the expected values come from the same symbols re-packed from symbol 0 on)
    L2, K2, first = 6_500_000, 63, 5
    seq2 = cap.Seq(words.ctypes.data, L2, first, 0, 4, 0)
    n2 = L2 - K2 + 1
    fw, rv = np.zeros((n2, 2), np.uint64), np.zeros((n2, 2), np.uint64)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq2), K2, 2, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
    nibbles = np.stack([(words >> np.uint64(4 * j)) & np.uint64(0xF) for j in range(16)], axis=1).reshape(-1)[first:first + L2]
    pad = np.concatenate([nibbles, np.zeros((-len(nibbles)) % 16 + 16, np.uint64)]).reshape(-1, 16)
    sub = np.zeros(len(pad), np.uint64)
    for j in range(16):
        sub |= pad[:, j] << np.uint64(4 * j)
    efw, erv, _ = orc.fwrv(sub, L2, 4, 2, K2)
    assert np.array_equal(fw, efw) and np.array_equal(rv, erv)
    # SpacedDNAMers{21,3}: 13 M elements of 8 B -> 104 MB -> four chunks
    L3, K3, J3 = 39_000_000, 21, 3
    w3 = orc.synth_words(78, 0, L3 // 16 + 2, 4)
    seq3, keep3 = make_seq(km, w3, L3, 4)
    n3 = (L3 - K3) // J3 + 1
    sp = np.zeros((n3, 1), np.uint64)
    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq3), K3, J3, 2, vp(sp), cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
    es, _ = orc.spaced(w3, L3, 4, 2, K3, J3)
    assert res.n_out == n3 and np.array_equal(sp, es)


def test_ascii_table_sources_on_strided_and_tuple_kernels(km, ctx, orc):
    """Text into a 4-bit alphabet goes through the alphabet's 256-entry table in LDS (stage_word), and the strided / tuple
    instantiations of the stream kernel stage their first tile before the tile loop's first barrier: the table must be complete
    before any wavefront looks a byte up (round 3 advisor: a missing barrier behind the table fill).  Many workgroups of 256
    threads, repeated: SpacedKmers with strides 2..32, Tuple{Kmer,Kmer} and Tuple{Kmer,UInt64} elements, against the oracle."""
    cap = km._capi
    rng = np.random.default_rng(4242)
    L = 400_000
    text = naive.random_text(rng, L, p_amb=0.05)
    text = "".join(c.lower() if rng.random() < 0.3 else c for c in text)
    seq, keep = ascii_seq(km, text, L)
    res = cap.Result()
    ctx.set_param(cap.PARAM_BLOCK_THREADS, 256)
    try:
        for rep in range(3):
            for K, J in ((5, 3), (21, 3), (16, 2), (7, 32)):
                n = (L - K) // J + 1
                N = (4 * K + 63) // 64
                out = np.zeros((n, N), np.uint64)
                assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 4, vp(out), 0, C.byref(res)) == 0, ctx.last_error()
                ek, er = orc.spaced(keep, L, 8, 4, K, J)
                assert er.status == 0 and np.array_equal(out, ek), (rep, K, J)
            for K in (9, 16, 31):
                n = L - K + 1
                N = (4 * K + 63) // 64
                t = np.zeros((n, 2 * N), np.uint64)
                assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 4, vp(t), None, cap.OUT_TUPLES, C.byref(res)) == 0, ctx.last_error()
                efw, erv, _ = orc.fwrv(keep, L, 8, 4, K)
                assert np.array_equal(t, np.concatenate([efw, erv], axis=1)), (rep, K)
                t = np.zeros((n, N + 1), np.uint64)
                assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 4, vp(t), None, 3, cap.OUT_TUPLES, C.byref(res)) == 0
                ek, eh, _ = orc.canonical(keep, L, 8, 4, K, seed=3)
                assert np.array_equal(t, np.concatenate([ek, eh[:, None]], axis=1)), (rep, K)
    finally:
        ctx.set_param(cap.PARAM_BLOCK_THREADS, 0)


def test_ascii_errors(km, ctx, orc):
    cap = km._capi
    rng = np.random.default_rng(89)
    bad_bytes = [ord("P"), ord("X"), ord("\n"), ord(" "), 0xC3, ord("U"), ord("n"), 0]
    for dst in (2, 4):
        for K in (3, 31):
            for L in (K, 300, 10000):
                for _ in range(4):
                    raw = bytearray(naive.random_text(rng, L).encode())
                    for p in rng.integers(0, L, size=2):
                        raw[p] = bad_bytes[int(rng.integers(0, len(bad_bytes)))]
                    seq, keep = ascii_seq(km, bytes(raw), L)
                    out = np.zeros((L, (K * dst + 63) // 64), np.uint64)
                    res = cap.Result()
                    rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(out), None, 0, C.byref(res))
                    _, er = orc.fw_kmers(keep, L, 8, dst, K)
                    assert (rc, res.err_pos, res.err_enc) == (er.status, er.err_pos, er.err_enc), (dst, K, L, bytes(raw[:40]))
                    # UnambiguousKmers: IUPAC letters are skipped, other bytes throw (common.jl:22-32)
                    if dst == 2:
                        rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, 0, C.byref(res))
                        _, _, er = orc.unambiguous(keep, L, 8, K)
                        assert (rc, res.err_pos if rc else 0, res.err_enc if rc else 0) == \
                            (er.status, er.err_pos, er.err_enc), (K, L)
    # a source shorter than K is still scanned by UnambiguousKmers (UnambiguousKmers.jl:117-123)
    seq, keep = ascii_seq(km, "AC!", 3)
    res = cap.Result()
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), 5, 1, None, None, 0, 0, C.byref(res)) == cap.E_ENCODE
    assert (res.err_pos, res.err_enc) == (3, ord("!"))
    seq, keep = ascii_seq(km, "ACN", 3)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), 5, 1, None, None, 0, 0, C.byref(res)) == 0
    assert res.n_out == 0


def test_ascii_unambiguous(km, ctx, orc):
    rng = np.random.default_rng(90)
    cap = km._capi
    for K in (3, 21, 33):
        for L in (K, 700, 40000):
            text = naive.random_text(rng, L, p_amb=0.05)
            text = "".join(c.lower() if rng.random() < 0.5 else c for c in text)
            seq, keep = ascii_seq(km, text, L)
            res = cap.Result()
            assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, 0, C.byref(res)) == 0
            n = int(res.n_out)
            N = (2 * K + 63) // 64
            kmers = np.zeros((max(n, 1), N), np.uint64)
            starts = np.zeros(max(n, 1), np.int64)
            assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, vp(kmers), vp(starts), n, 0, C.byref(res)) == 0
            ek, es, _ = orc.unambiguous(keep, L, 8, K)
            assert np.array_equal(kmers[:n], ek) and np.array_equal(starts[:n], es)


def test_ascii_unambiguous_clean_text_of_the_other_alphabet(km, ctx, orc):
    """UnambiguousKmers reads bytes through ASCII_SKIPPING_LUT, in which T and U are the same symbol for DNA and RNA kmers
    alike (src/iterators/common.jl:22-32): UnambiguousDNAMers over clean "ACGU" text and UnambiguousRNAMers over clean
    "ACGT" text keep EVERY window (the all-kept case), and leave no error behind for the next call on the context."""
    rng = np.random.default_rng(91)
    cap = km._capi
    for alphabet, letters in ((0, "ACGU"), (1, "ACGT"), (0, "acgu"), (1, "ACGTacgtUu")):
        for K in (4, 31, 33):
            for L in (K, 5000, 70001):
                text = "".join(letters[i] for i in rng.integers(0, len(letters), L))
                seq, keep = ascii_seq(km, text, L, alphabet=alphabet)
                res = cap.Result()
                assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, 0, C.byref(res)) == 0
                n = int(res.n_out)
                assert n == L - K + 1
                N = (2 * K + 63) // 64
                kmers = np.zeros((n, N), np.uint64)
                starts = np.zeros(n, np.int64)
                assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, vp(kmers), vp(starts), n, 0, C.byref(res)) == 0
                ek, es, _ = orc.unambiguous(keep, L, 8, K)
                assert np.array_equal(kmers, ek) and np.array_equal(starts, es), (alphabet, letters, K, L)
                # the context's error slot must still be clean: a strict call on valid input succeeds
                ok_text = "ACGT" * 20
                seq2, keep2 = ascii_seq(km, ok_text, len(ok_text), alphabet=0)
                out = np.zeros((len(ok_text) - 3, 1), np.uint64)
                assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq2), 4, 2, vp(out), None, 0, C.byref(res)) == 0, (res.status, res.err_pos)


def test_ascii_every_byte_value(km, ctx, orc):
    """Text into a 2-bit alphabet is recoded arithmetically, eight bytes at a time (stage_word): every one of the 256 byte values,
    at every position of a source word, must give what BioSequences.ascii_encode gives -- the symbol, or the EncodeError with that
    byte -- for DNA and for RNA kmers, through the stride-1, the strided and the forward-only kernels."""
    cap = km._capi
    base = b"ACGTTGCAAGGCTTACGATCGATTAGC"
    for alphabet in (0, 1):
        b0 = base.replace(b"T", b"U") if alphabet else base
        for value in range(256):
            slot = value % 8 + 8                       # walk the byte through the positions of a word
            raw = b0[:slot] + bytes([value]) + b0[slot + 1:]
            L = len(raw)
            words = naive.ascii_words(raw)
            seq = cap.Seq(words.ctypes.data, L, 0, 0, 8, alphabet)
            osrc = 9 if alphabet else 8
            for K in (3, 5):
                n = L - K + 1
                res = cap.Result()
                ck, hs = np.zeros((n, 1), np.uint64), np.zeros(n, np.uint64)
                rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, vp(ck), vp(hs), 0, cap.MEM_HOST, C.byref(res))
                ek, eh, eres = orc.canonical(words, L, osrc, 2, K)
                assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (alphabet, value, K)
                if rc == 0:
                    assert np.array_equal(ck, ek) and np.array_equal(hs, eh), (alphabet, value)
                fw = np.zeros((n, 1), np.uint64)
                rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, vp(fw), None, cap.MEM_HOST, C.byref(res))   # forward-only kernel
                efw, eres = orc.fw_kmers(words, L, osrc, 2, K)
                assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (alphabet, value, K)
                if rc == 0:
                    assert np.array_equal(fw, efw)
                m = (L - K) // 2 + 1
                sp = np.zeros((m, 1), np.uint64)
                rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, 2, 2, vp(sp), cap.MEM_HOST, C.byref(res))
                es, eres = orc.spaced(words, L, osrc, 2, K, 2)
                assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (alphabet, value, K)
                if rc == 0:
                    assert np.array_equal(sp, es)


def test_symbol_vector_sources(km, ctx, orc):
    """GenericRecoding sources (src/construction.jl:90-98; FwKmers.jl:80-86, CanonicalKmers.jl:81-91,
    construction_utils.jl:90-103, :161-172): a Vector{DNA} / Vector{RNA} -- one BioSymbols value per byte
    (kmers_seq.alphabet = KMERS_ALPHABET_SYMBOLS).  Same kmers and the same EncodeError (position, symbol) as the oracle;
    UnambiguousKmers has no method for such a source."""
    cap = km._capi
    rng = np.random.default_rng(2027)
    enc4 = {c: i for i, c in enumerate("-ACMGRSVTWYHKDBN")}
    for dst in (2, 4):
        for K in (1, 5, 31, 33, 64, 70 if dst == 2 else 40):
            N = (K * dst + 63) // 64
            for L in (K, K + 13, 5000, 40013):
                for p_amb in (0.0, 0.001):
                    text = naive.random_text(rng, L, p_amb=p_amb)
                    raw = bytes(enc4[c] for c in text)
                    if p_amb and L > 100 and dst == 4:
                        raw = raw[:77] + bytes([0x41]) + raw[78:]   # not a nucleotide value: an error for both widths
                    words = naive.ascii_words(raw)
                    seq = cap.Seq(words.ctypes.data, L, 0, 0, 8, 2)
                    n = L - K + 1
                    res = cap.Result()
                    fw, rv = np.zeros((n, N), np.uint64), np.zeros((n, N), np.uint64)
                    rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res))
                    efw, erv, eres = orc.fwrv(words, L, 10, dst, K)
                    assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (dst, K, L)
                    if rc == 0:
                        assert np.array_equal(fw, efw) and np.array_equal(rv, erv), (dst, K, L)
                    ck, hs = np.zeros((n, N), np.uint64), np.zeros(n, np.uint64)
                    rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(ck), vp(hs), 9, cap.MEM_HOST, C.byref(res))
                    ek, eh, eres = orc.canonical(words, L, 10, dst, K, seed=9)
                    assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc)
                    if rc == 0:
                        assert np.array_equal(ck, ek) and np.array_equal(hs, eh)
                    for J in (3, K + 3):
                        m = (L - K) // J + 1
                        out = np.zeros((m, N), np.uint64)
                        rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(out), cap.MEM_HOST, C.byref(res))
                        es, eres = orc.spaced(words, L, 10, dst, K, J)
                        assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (dst, K, L, J)
                        if rc == 0:
                            assert np.array_equal(out, es)
    # UnambiguousKmers over a collection of symbols: the reference's generic method (UnambiguousKmers.jl:88-106; its test
    # runtests.jl:834-840).  Ambiguous symbols are skipped; the gap is an EncodeError (a four-bit sequence skips it).
    enc4 = {c: i for i, c in enumerate("-ACMGRSVTWYHKDBN")}
    for L in (1, 30, 1000, 70_001):
        for p_amb, gaps in ((0.0, 0), (0.03, 0), (0.03, 2)):
            text = list(naive.random_text(rng, L, p_amb=p_amb).replace("-", "N"))
            for pos in rng.integers(0, L, gaps):
                text[int(pos)] = "-"
            words = naive.ascii_words(bytes(enc4[c] for c in text))
            seq = cap.Seq(words.ctypes.data, L, 0, 0, 8, 2)
            for K, J in ((1, 1), (4, 1), (31, 1), (33, 1), (140, 1), (21, 3)):
                res = cap.Result()
                ek, es, eres = orc.unambiguous(words, L, 10, K)
                rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, None, None, 0, cap.MEM_HOST, C.byref(res))
                assert (rc, res.err_pos, res.err_enc) == (eres.status, eres.err_pos, eres.err_enc), (L, K, J, gaps)
                if rc == 0:
                    keep = (es - 1) % J == 0
                    m = int(keep.sum())
                    assert res.n_out == m
                    kmers, starts = np.zeros((max(m, 1), (2 * K + 63) // 64), np.uint64), np.zeros(max(m, 1), np.int64)
                    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, J, vp(kmers), vp(starts), m, cap.MEM_HOST, C.byref(res)) == 0
                    assert np.array_equal(kmers[:m], ek[keep]) and np.array_equal(starts[:m], es[keep]), (L, K, J)
                    if K <= 64 and J == 1:  # the fused XOR reducer over the same iterator
                        val = C.c_uint64()
                        assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_UNAMBIGUOUS, 1, C.byref(val), 0, C.byref(res)) == 0
                        assert val.value == (int(np.bitwise_xor.reduce(ek[:, 0])) if len(ek) else 0), (L, K)
    res = cap.Result()
    seq = cap.Seq(naive.ascii_words(b"ACGT").ctypes.data, 4, 0, 0, 8, 3)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 2, 2, None, None, 0, C.byref(res)) == cap.E_BADARG


def test_offset_views(km, ctx, orc):
    """first_base != 0 (LongSubSeq / halo shards): same result as the oracle on the re-packed view."""
    cap = km._capi
    rng = np.random.default_rng(3)
    text = naive.random_text(rng, 9000)
    for bits in (2, 4):
        words = naive.longseq_words(text, bits)
        for first in (0, 1, 15, 16, 17, 31, 33, 1000, 4097):
            for K in (3, 31, 33):
                L = len(text) - first - 5
                rc, kmers, hashes, res = run_canonical(km, ctx, words, L, bits, K, 7, first_base=first)
                assert rc == 0
                sub = naive.longseq_words(text[first:first + L], bits)
                ek, eh, _ = orc.canonical(sub, L, bits, 2, K, seed=7)
                assert np.array_equal(kmers, ek), (bits, first, K)
                assert np.array_equal(hashes, eh)


def test_symbols_outside_a_view_do_not_touch_its_kmers(km, ctx, orc):
    """A view (first_base != 0, or a length that ends inside a word) shares its first and last source word with symbols that are
    not part of it.  Whatever those are -- every one of the 16 values of a LongDNA{4} nibble, gap and ambiguity codes included --
    the kmers of the view are the kmers of the view (found by the extended fuzz, seed 5071: the 4-bit staging let the code of
    a nibble that is not one-hot spill into its neighbour's field)."""
    cap = km._capi
    rng = np.random.default_rng(5071)
    alphabet = "-ACMGRSVTWYHKDBN"  # nibble value = index
    for K, L, first in ((32, 33, 63), (31, 40, 15), (5, 20, 17), (32, 64, 1), (17, 100, 48)):
        clean = naive.random_text(rng, first + L + 40)
        for v in range(16):
            for where in ("before", "after", "both"):
                t = list(clean)
                if where in ("before", "both"):
                    t[first - 1] = alphabet[v]
                if where in ("after", "both"):
                    t[first + L] = alphabet[v]
                text = "".join(t)
                words = naive.longseq_words(text, 4)
                sub = naive.longseq_words(text[first:first + L], 4)
                rc, kmers, hashes, res = run_canonical(km, ctx, words, L, 4, K, 3, first_base=first)
                ek, eh, _ = orc.canonical(sub, L, 4, 2, K, seed=3)
                assert rc == 0 and np.array_equal(kmers, ek) and np.array_equal(hashes, eh), (K, L, first, v, where)
                seq, keep = make_seq(km, words, L, 4, first)
                val, r2 = C.c_uint64(), cap.Result()
                assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), cap.MEM_HOST, C.byref(r2)) == 0
                assert val.value == orc.reduce_xor_canonical(sub, L, 4, 2, K)[0], (K, L, first, v, where)


def test_encode_errors_match_oracle(km, ctx, orc):
    """First offending symbol (position and raw encoding) exactly as the reference would throw
    (construction.jl:108-110; raise sites FwKmers.jl:112, CanonicalKmers.jl:139)."""
    cap = km._capi
    rng = np.random.default_rng(21)
    for K in (1, 3, 31, 33, 65, 128):
        for L in (K, 500, 20000):
            for trial in range(4):
                text = list(naive.random_text(rng, L))
                nbad = int(rng.integers(1, 4))
                for p in rng.integers(0, L, size=nbad):
                    text[p] = "NMRWSYKVHDB-"[int(rng.integers(0, 12))]
                text = "".join(text)
                words = naive.longseq_words(text, 4)
                rc, _, _, res = run_canonical(km, ctx, words, L, 4, K)
                _, _, eres = orc.canonical(words, L, 4, 2, K)
                assert rc == cap.E_ENCODE and eres.status == 1
                assert (res.err_pos, res.err_enc) == (eres.err_pos, eres.err_enc), (K, L, text[:40])
                # the context is usable again and clean inputs pass
                rc, kmers, _, res = run_canonical(km, ctx, naive.longseq_words("ACGT" * 20, 4), 80, 4, min(K, 31))
                assert rc == 0 and res.status == 0
    # garbage beyond `len` in the last word is not inspected; L < K inspects nothing (FwKmers.jl:63)
    words = naive.longseq_words("ACGTACGTAC" + "N" * 6, 4)
    rc, kmers, _, res = run_canonical(km, ctx, words, 10, 4, 4)
    assert rc == 0 and len(kmers) == 7
    rc, kmers, _, res = run_canonical(km, ctx, naive.longseq_words("NN", 4), 2, 4, 3)
    assert rc == 0 and res.n_out == 0


@pytest.mark.parametrize("bits", [2, 4])
def test_spaced_parity(km, ctx, orc, bits):
    """SpacedKmers{A,K,J} (SpacedKmers.jl:38-42, :92-139): tile path (J <= 32) and gather path."""
    cap = km._capi
    for K, J in [(3, 2), (2, 4), (3, 3), (4, 3), (21, 3), (31, 7), (33, 5), (64, 32), (5, 40), (31, 31), (31, 33),
                 (63, 100)]:
        for L in (K - 1, K, K + J, 10 * K + 3 * J + 1, 30011):
            words = orc.synth_words(5 + J, 1, (L * bits + 63) // 64 + 1, bits)
            n = 0 if L < K else (L - K) // J + 1
            N = (2 * K + 63) // 64
            out = np.zeros((max(n, 1), N), dtype=np.uint64)
            res = cap.Result()
            seq, keep = make_seq(km, words, L, bits)
            rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, vp(out), cap.MEM_HOST, C.byref(res))
            assert rc == 0, (K, J, L, ctx.last_error())
            ek, eres = orc.spaced(words, L, bits, 2, K, J)
            assert res.n_out == n == len(ek)
            assert np.array_equal(out[:n], ek), (K, J, L)


def test_spaced_error_semantics(km, ctx, orc):
    """J < K: every symbol up to the end of the last kmer is inspected; J >= K: gaps are not
    (SpacedKmers.jl:133-137; test/runtests.jl:868-869)."""
    cap = km._capi
    rng = np.random.default_rng(17)
    for K, J in [(3, 2), (21, 3), (3, 4), (5, 9), (31, 31), (4, 50)]:
        for L in (K, K + J, 200, 9000):
            for p in (0.002, 0.05):
                text = naive.random_text(rng, L, p_amb=p)
                words = naive.longseq_words(text, 4)
                n = 0 if L < K else (L - K) // J + 1
                out = np.zeros((max(n, 1), 1), dtype=np.uint64)
                res = cap.Result()
                seq, keep = make_seq(km, words, L, 4)
                rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, vp(out), cap.MEM_HOST, C.byref(res))
                ek, eres = orc.spaced(words, L, 4, 2, K, J)
                if eres.status == 0:
                    assert rc == 0 and np.array_equal(out[:n], ek)
                else:
                    assert rc == cap.E_ENCODE
                    assert (res.err_pos, res.err_enc) == (eres.err_pos, eres.err_enc), (K, J, L)
    # the reference's own case: SpacedDNAMers{3,4}("TAGAWWWW") throws at W (position 5)
    words = naive.longseq_words("TAGAWWWW", 4)
    seq, keep = make_seq(km, words, 8, 4)
    out = np.zeros((2, 1), dtype=np.uint64)
    res = cap.Result()
    rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), 3, 4, 2, vp(out), cap.MEM_HOST, C.byref(res))
    assert rc == cap.E_ENCODE and res.err_pos == 5 and res.err_enc == naive.DNA4["W"]


def run_unambiguous(km, ctx, words, L, bits, K, stride=1, origin=0):
    cap = km._capi
    res = cap.Result()
    seq, keep = make_seq(km, words, L, bits, 0, origin)
    rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, stride, None, None, 0, cap.MEM_HOST, C.byref(res))
    assert rc == 0
    n = int(res.n_out)
    N = (2 * K + 63) // 64
    kmers = np.zeros((max(n, 1), N), dtype=np.uint64)
    starts = np.zeros(max(n, 1), dtype=np.int64)
    rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, stride, vp(kmers), vp(starts), n, cap.MEM_HOST,
                                   C.byref(res))
    assert rc == 0 and res.n_out == n
    return kmers[:n], starts[:n]


def test_unambiguous_device_outputs_any_alignment(km, ctx, orc):
    """Device-resident outputs, dense and sparse survivors: mostly clean sequence takes the kernel variant
    whose all-kept wavefronts skip compaction and store 16 bytes per lane (16-byte aligned outputs only);
    buffers that are only 8-byte aligned, and sparse survivors, take the compacting path."""
    cap = km._capi
    L, K = 400_003, 21
    for amb in (0, 2, 2621):  # p(N) = 0, 3e-5 (99.9 % of the starts kept), 0.04
        words = orc.synth_words(21 + amb, 0, L // 16 + 1, 4, amb)
        ek, es, _ = orc.unambiguous(words, L, 4, K)
        n = len(ek)
        d_src = ctx.alloc(len(words) * 8 + 16)
        ctx.h2d(d_src, words)
        seq = cap.Seq(d_src, L, 0, 0, 4, 0)
        res = cap.Result()
        d_k, d_s = ctx.alloc(n * 8 + 32), ctx.alloc(n * 8 + 32)
        for shift in (0, 8):
            rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, d_k + shift, d_s + shift, n, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0 and res.n_out == n, ctx.last_error()
            kmers, starts = np.zeros(n, np.uint64), np.zeros(n, np.int64)
            ctx.d2h(kmers, d_k + shift)
            ctx.d2h(starts, d_s + shift)
            assert np.array_equal(kmers, ek[:, 0]) and np.array_equal(starts, es), (amb, shift)
        for d in (d_src, d_k, d_s):
            ctx.free(d)


def test_unambiguous_parity(km, ctx, orc):
    """UnambiguousKmers (UnambiguousKmers.jl:64-77, :134-148): windows and 1-based starts."""
    rng = np.random.default_rng(31)
    for K in (1, 3, 21, 31, 32, 33, 64, 65, 96, 97, 128):
        for L in (K - 1, K, 300, 5000, 33000):
            if L < 0:
                continue
            for p in (0.0, 0.04, 0.4):
                text = naive.random_text(rng, L, p_amb=p)
                words = naive.longseq_words(text, 4)
                kmers, starts = run_unambiguous(km, ctx, words, L, 4, K)
                ek, es, _ = orc.unambiguous(words, L, 4, K)
                assert np.array_equal(kmers, ek), (K, L, p)
                assert np.array_equal(starts, es), (K, L, p)
            text = naive.random_text(rng, L)
            words = naive.longseq_words(text, 2)
            kmers, starts = run_unambiguous(km, ctx, words, L, 2, K)
            ek, es, _ = orc.unambiguous(words, L, 2, K)
            assert np.array_equal(kmers, ek) and np.array_equal(starts, es)


def test_unambiguous_single_pass_device_path(km, ctx, orc):
    """The one-pass kernel proper (device outputs: no count pass): tile descriptors + look-back across many tiles, every
    tile size, workgroups that draw several tickets (capped grid), kmers of one to four words, stride lattices, and the
    capacity contract: nothing is written at or beyond `capacity`, KMERS_E_CAPACITY reports the count needed."""
    cap = km._capi
    rng = np.random.default_rng(131)
    L = 300_017
    guard = np.uint64(0xDEADBEEFDEADBEEF)
    for p_amb in (0.04, 0.0005, 0.5):
        text = naive.random_text(rng, L, p_amb=p_amb)
        words = naive.longseq_words(text, 4)
        d_src = ctx.alloc(words.nbytes + 16)
        ctx.h2d(d_src, words)
        for K, stride in ((21, 1), (31, 1), (33, 1), (65, 1), (97, 1), (128, 1), (21, 3), (70, 5)):
            ek, es, _ = orc.unambiguous(words, L, 4, K)
            keep = (es - 1) % stride == 0
            ek, es = ek[keep], es[keep]
            n, N = len(ek), (2 * K + 63) // 64
            seq = cap.Seq(d_src, L, 0, 1000, 4, 0)   # index_origin: starts come out global
            for tile, grid in ((0, 0), (1024, 0), (4096, 7), (32768, 2), (2048, 300)):
                ctx.set_param(cap.PARAM_TILE_KMERS, tile)
                ctx.set_param(cap.PARAM_MAX_GRID, grid)
                for capacity in (n, n + 1000, max(n - 1, 0) if n else 0, n // 2):
                    room = max(n, 1) + 1000 + 8
                    hk = np.full((room, N), guard, np.uint64)
                    hs = np.full(room, guard.view(np.int64), np.int64)
                    dk, ds = ctx.alloc(hk.nbytes), ctx.alloc(hs.nbytes)
                    ctx.h2d(dk, hk)
                    ctx.h2d(ds, hs)
                    res = cap.Result()
                    rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, stride, dk, ds, capacity, cap.MEM_DEVICE, C.byref(res))
                    ctx.d2h(hk, dk)
                    ctx.d2h(hs, ds)
                    ctx.free(dk)
                    ctx.free(ds)
                    assert res.n_out == n, (K, stride, tile, grid, capacity, res.n_out, n)
                    assert np.all(hk[capacity:] == guard) and np.all(hs[capacity:] == guard.view(np.int64))   # never beyond the capacity
                    if capacity >= n:
                        assert rc == 0
                        assert np.array_equal(hk[:n], ek) and np.array_equal(hs[:n], es + 1000), (K, stride, tile, grid, p_amb)
                    else:
                        assert rc == cap.E_CAPACITY and res.status == cap.E_CAPACITY
            ctx.set_param(cap.PARAM_TILE_KMERS, 0)
            ctx.set_param(cap.PARAM_MAX_GRID, 0)
        ctx.free(d_src)


def test_unambiguous_asynchronous_form(km, ctx, orc):
    """KMERS_MEM_DEVICE | KMERS_ASYNC: the one pass is only enqueued; kmers_sync reports the count (and KMERS_E_CAPACITY with the
    count needed when the outputs were too small -- nothing stored beyond them) and an EncodeError of a byte source; of several
    asynchronous calls before one sync the last one's count is reported.  A 2-bit source knows its count at once."""
    cap = km._capi
    rng = np.random.default_rng(77)
    L, K = 200_003, 31
    text = naive.random_text(rng, L, p_amb=0.03)
    words = naive.longseq_words(text, 4)
    ek, es, _ = orc.unambiguous(words, L, 4, K)
    n = len(ek)
    d_src = ctx.alloc(words.nbytes + 16)
    ctx.h2d(d_src, words)
    seq = cap.Seq(d_src, L, 0, 0, 4, 0)
    guard = np.uint64(0xDEADBEEFDEADBEEF)
    ASYNC = cap.MEM_DEVICE | cap.ASYNC
    for capacity in (n + 5, n, n - 1):
        hk = np.full(n + 64, guard, np.uint64)
        hs = np.full(n + 64, guard.view(np.int64), np.int64)
        dk, ds = ctx.alloc(hk.nbytes), ctx.alloc(hs.nbytes)
        ctx.h2d(dk, hk)
        ctx.h2d(ds, hs)
        res = cap.Result()
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, dk, ds, capacity, ASYNC, C.byref(res)) == 0 and res.n_out == 0
        # a second one behind it (same outputs: the sync reports the last one's count), and other asynchronous work
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, dk, ds, capacity, ASYNC, C.byref(res)) == 0
        d_tmp = ctx.alloc(8 * L)
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), 5, 4, d_tmp, None, 0, ASYNC, C.byref(res)) == 0
        rc, sres = ctx.sync()
        ctx.d2h(hk, dk)
        ctx.d2h(hs, ds)
        assert sres.n_out == n, (capacity, sres.n_out, n)
        assert np.all(hs[capacity:] == guard.view(np.int64))
        if capacity >= n:
            assert rc == 0 and np.array_equal(hk[:n], ek[:, 0]) and np.array_equal(hs[:n], es)
        else:
            assert rc == cap.E_CAPACITY and sres.status == cap.E_CAPACITY
        rc, sres = ctx.sync()          # nothing pending any more
        assert rc == 0 and sres.n_out == 0
        ctx.free(dk)
        ctx.free(ds)
        ctx.free(d_tmp)
    # host pointers and the size query have no asynchronous form
    res = cap.Result()
    out = np.zeros(n + 1, np.uint64)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, vp(out), None, n, cap.ASYNC, C.byref(res)) == cap.E_BADARG
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, ASYNC, C.byref(res)) == cap.E_BADARG
    # text with a byte that is no nucleotide: the sync reports the EncodeError
    raw = bytearray(naive.random_text(rng, 5000).encode())
    raw[3210] = ord("!")
    wb = naive.ascii_words(bytes(raw))
    d_txt = ctx.alloc(wb.nbytes + 16)
    ctx.h2d(d_txt, wb)
    seqt = cap.Seq(d_txt, len(raw), 0, 0, 8, 0)
    dk, ds = ctx.alloc(10_000 * 8), ctx.alloc(10_000 * 8)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seqt), 21, 1, dk, ds, 5000, ASYNC, C.byref(res)) == 0
    rc, sres = ctx.sync()
    assert rc == cap.E_ENCODE and sres.err_pos == 3211 and sres.err_enc == ord("!")
    # a 2-bit source: nothing can be dropped, the count is known at the call
    w2 = naive.longseq_words(naive.random_text(rng, 10_000), 2)
    d2 = ctx.alloc(w2.nbytes + 16)
    ctx.h2d(d2, w2)
    seq2 = cap.Seq(d2, 10_000, 0, 0, 2, 0)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq2), 21, 1, dk, ds, 10_000, ASYNC, C.byref(res)) == 0 and res.n_out == 9980
    rc, sres = ctx.sync()
    assert rc == 0
    e2, s2, _ = orc.unambiguous(w2, 10_000, 2, 21)
    hk = np.zeros(9980, np.uint64)
    ctx.d2h(hk, dk)
    assert np.array_equal(hk, e2[:, 0])
    for p in (d_src, d_txt, d2, dk, ds):
        ctx.free(p)


def test_unambiguous_kmers_wider_than_four_words(km, ctx, orc):
    """UnambiguousKmers{A,K} with K > 128 (src/kmer.jl:97-111 puts no bound on N): the single-pass kernel's run-time-width
    instantiation.  Against the oracle up to its 8 words, against the naive slicer beyond; host and device paths, a stride
    lattice, several tiles."""
    cap = km._capi
    rng = np.random.default_rng(777)
    for K, L, p_amb in ((129, 40_000, 0.002), (200, 70_001, 0.003), (256, 9_000, 0.0), (130, 300, 0.0), (256, 255, 0.0)):
        text = naive.random_text(rng, L, p_amb=p_amb)
        words = naive.longseq_words(text, 4)
        ek, es, _ = orc.unambiguous(words, L, 4, K)
        kmers, starts = run_unambiguous(km, ctx, words, L, 4, K)                 # host path (count, then emit)
        assert np.array_equal(kmers, ek) and np.array_equal(starts, es), (K, L)
        kmers, starts = run_unambiguous(km, ctx, words, L, 4, K, stride=5)
        keep = (es - 1) % 5 == 0
        assert np.array_equal(kmers, ek[keep]) and np.array_equal(starts, es[keep]), (K, L)
        # device path (one pass), small tiles
        n, N = len(ek), (2 * K + 63) // 64
        d_src = ctx.alloc(words.nbytes + 16)
        ctx.h2d(d_src, words)
        for tile in (0, 1024):
            ctx.set_param(cap.PARAM_TILE_KMERS, tile)
            hk, hs = np.zeros((n + 8, N), np.uint64), np.zeros(n + 8, np.int64)
            dk, ds = ctx.alloc(hk.nbytes), ctx.alloc(hs.nbytes)
            res = cap.Result()
            seq = cap.Seq(d_src, L, 0, 0, 4, 0)
            assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, dk, ds, n + 8, cap.MEM_DEVICE, C.byref(res)) == 0
            assert res.n_out == n
            ctx.d2h(hk, dk)
            ctx.d2h(hs, ds)
            ctx.free(dk)
            ctx.free(ds)
            assert np.array_equal(hk[:n], ek) and np.array_equal(hs[:n], es), (K, L, tile)
        ctx.set_param(cap.PARAM_TILE_KMERS, 0)
        ctx.free(d_src)
    # a LongDNA{2} or text source with K > 128 (nothing ambiguous in a 2-bit source: still the run-time-width kernel, not the
    # four-word stream kernel -- the round-2 fuzz found that dispatch returning unwritten kmers)
    for src, K, L in ((2, 129, 129), (2, 200, 5003), (2, 256, 257), (8, 150, 2000)):
        text = naive.random_text(rng, L, p_amb=0.0 if src == 2 else 0.004)
        words = naive.longseq_words(text, 2) if src == 2 else naive.ascii_words(text)
        ek, es, _ = orc.unambiguous(words, L, src, K)
        for stride in (1, 3):
            kmers, starts = run_unambiguous(km, ctx, words, L, src, K, stride=stride)
            keep = (es - 1) % stride == 0
            assert np.array_equal(kmers, ek[keep]) and np.array_equal(starts, es[keep]), (src, K, L, stride)
    # beyond the oracle's widths: K = 700 (22 words) and K = 3000 over text with a few ambiguity codes, against the naive slicer
    for K, L in ((700, 4000), (3000, 12_000)):
        text = list(naive.random_text(rng, L))
        for pos in rng.integers(0, L, 3):
            text[int(pos)] = "N"
        text = "".join(text)
        kmers, starts = run_unambiguous(km, ctx, naive.longseq_words(text, 4), L, 4, K)
        exp = naive.unambiguous(text, K)
        assert [tuple(int(x) for x in r) for r in kmers] == [k_ for k_, _ in exp] and list(starts) == [i for _, i in exp], K
    res = cap.Result()
    seq, keep = make_seq(km, naive.longseq_words("ACGT" * 10, 4), 40, 4)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), 31000, 1, None, None, 0, 0, C.byref(res)) == cap.E_UNSUPPORTED


def test_spaced_skip_variant(km, ctx, orc):
    """BASELINE.json config 5 "with ambiguous-base skip" = the elements (kmer, i) of
    UnambiguousDNAMers{K}(seq) with (i-1) % J == 0 (docs/src/faq.md:28-33)."""
    rng = np.random.default_rng(37)
    for K, J in [(21, 3), (5, 2), (31, 7), (4, 1000), (3, 32767), (3, 40000)]:
        L = 20000 if J < 1000 else 200000
        text = naive.random_text(rng, L, p_amb=0.04)
        words = naive.longseq_words(text, 4)
        kmers, starts = run_unambiguous(km, ctx, words, L, 4, K, stride=J)
        ek, es, _ = orc.unambiguous(words, L, 4, K)
        keep = (es - 1) % J == 0
        assert np.array_equal(kmers, ek[keep]) and np.array_equal(starts, es[keep])
        # equals strict Spaced output on windows without ambiguity
        clean = "".join(c if c in "ACGT" else "A" for c in text)
        exp = [w for w, i in zip(naive.spaced(clean, K, J, 2), range(0, L, J))
               if naive.is_certain(text[i:i + K])]
        assert [tuple(int(x) for x in r) for r in kmers] == exp


def test_capacity_error(km, ctx):
    cap = km._capi
    words = naive.longseq_words("ACGTACGTACGT", 4)
    seq, keep = make_seq(km, words, 12, 4)
    kmers = np.zeros((2, 1), dtype=np.uint64)
    starts = np.zeros(2, dtype=np.int64)
    res = cap.Result()
    rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), 3, 1, vp(kmers), vp(starts), 2, cap.MEM_HOST,
                                   C.byref(res))
    assert rc == cap.E_CAPACITY and res.n_out == 10


def test_reduce_xor(km, ctx, orc):
    """Fused consumer of test/benchmark.jl:9-15."""
    cap = km._capi
    for bits in (2, 4):
        for K in (7, 31, 33):
            L = 100003
            words = orc.synth_words(77, 0, (L * bits + 63) // 64 + 1, bits)
            seq, keep = make_seq(km, words, L, bits)
            val = C.c_uint64()
            res = cap.Result()
            rc = ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), cap.MEM_HOST,
                                          C.byref(res))
            assert rc == 0
            exp, _ = orc.reduce_xor_canonical(words, L, bits, 2, K)
            assert val.value == exp
            rc = ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 0, C.byref(val), cap.MEM_HOST,
                                          C.byref(res))
            fw, _ = orc.fw_kmers(words, L, bits, 2, K)
            assert rc == 0 and val.value == int(np.bitwise_xor.reduce(fw[:, 0]))


def test_reduce_xor_every_k_sources_views_and_run_edges(km, ctx, orc):
    """The fused XOR reducer cuts a full run of 32 one-word kmers as windows of two 128-bit streams (run_kernel.hpp) and rolls
    the last, short run and two-word kmers: every K of both paths, every source width, views that start inside a word, and
    lengths that end a run / a tile (253 runs of 32 kmers) exactly, one kmer early and one kmer late."""
    cap = km._capi
    rng = np.random.default_rng(4242)
    tile = 253 * 32
    text = naive.random_text(rng, 3 * tile + 200)
    sources = {2: naive.longseq_words(text, 2), 4: naive.longseq_words(text, 4), 8: naive.ascii_words(text)}

    def check(bits, first, L, K):
        seq, keep = make_seq(km, sources[bits], L, bits, first)
        sub = text[first:first + L]
        ow = naive.ascii_words(sub) if bits == 8 else naive.longseq_words(sub if sub else "A", bits)
        val, res = C.c_uint64(), cap.Result()
        for canonical in (1, 0):
            rc = ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, canonical, C.byref(val), cap.MEM_HOST, C.byref(res))
            assert rc == 0, (bits, first, L, K, ctx.last_error())
            if canonical:
                exp, _ = orc.reduce_xor_canonical(ow, L, bits, 2, K)
            else:
                fw, _ = orc.fw_kmers(ow, L, bits, 2, K)
                exp = int(np.bitwise_xor.reduce(fw[:, 0])) if len(fw) else 0
            assert val.value == exp, (bits, first, L, K, canonical)

    for bits in (2, 4, 8):
        for K in list(range(1, 34)) + [47, 64]:
            check(bits, 0, 5000 + K, K)
        for first in (1, 7, 15, 16, 31, 33, 63):
            for K in (1, 16, 17, 31, 32):
                check(bits, first, 9000, K)
        for K in (5, 31, 32):
            for n_kmers in (1, 31, 32, 33, 63, 64, 65, tile - 1, tile, tile + 1, tile + 32, 2 * tile, 2 * tile + 31, 3 * tile):
                check(bits, 3, n_kmers + K - 1, K)
        check(bits, 0, 3, 5)  # shorter than a kmer: nothing to reduce


def test_batch_fx_hash_and_transforms(km, ctx, orc):
    """fx_hash and reverse/complement/reverse_complement/canonical/iscanonical on kmer arrays
    (kmer.jl:255-261; transformations.jl:1-41)."""
    cap = km._capi
    rng = np.random.default_rng(41)
    for bits in (2, 4):
        for K in (1, 7, 16, 31, 32, 33, 47, 63, 64, 96, 128):
            N = (K * bits + 63) // 64
            if N > 4:
                continue
            n = 300
            texts = [naive.random_text(rng, K, p_amb=0.2 if bits == 4 else 0.0) for _ in range(n)]
            arr = np.array([naive.kmer_words(t, bits) for t in texts], dtype=np.uint64).reshape(n, N)
            out = np.zeros(n, dtype=np.uint64)
            rc = ctx.lib.kmers_fx_hash(ctx.handle, vp(arr), N, n, 99, vp(out), cap.MEM_HOST)
            assert rc == 0
            assert out.tolist() == [naive.fx_hash(r, 99) for r in arr.tolist()]
            for op, fn in ((cap.OP_REVERSE, orc.reverse), (cap.OP_COMPLEMENT, orc.complement),
                           (cap.OP_REVCOMP, orc.reverse_complement), (cap.OP_CANONICAL, orc.canonical_kmer)):
                res = np.zeros((n, N), dtype=np.uint64)
                rc = ctx.lib.kmers_transform(ctx.handle, op, vp(arr), K, bits, n, vp(res), cap.MEM_HOST)
                assert rc == 0
                exp = [fn(tuple(r), K, bits) for r in arr.tolist()]
                assert [tuple(r) for r in res.tolist()] == exp, (bits, K, op)
            ls = np.zeros((n, N), dtype=np.uint64)
            rc = ctx.lib.kmers_transform(ctx.handle, cap.OP_TO_LONGSEQ, vp(arr), K, bits, n, vp(ls), cap.MEM_HOST)
            assert rc == 0
            assert [list(r) for r in ls.tolist()] == [list(naive.longseq_words(t, bits)[:N]) for t in texts], (bits, K)
            if bits == 2:
                gc = np.zeros(n, dtype=np.uint64)
                rc = ctx.lib.kmers_transform(ctx.handle, cap.OP_COUNT_GC, vp(arr), K, bits, n, vp(gc), cap.MEM_HOST)
                assert rc == 0 and gc.tolist() == [sum(c in "GC" for c in t) for t in texts]
            flags = np.zeros(n, dtype=np.uint64)
            rc = ctx.lib.kmers_transform(ctx.handle, cap.OP_ISCANONICAL, vp(arr), K, bits, n, vp(flags), cap.MEM_HOST)
            assert rc == 0
            assert flags.astype(bool).tolist() == [orc.iscanonical(tuple(r), K, bits) for r in arr.tolist()]


def test_synth_generator_matches_oracle(km, ctx, orc):
    for bits in (2, 4):
        for amb in (0, 2621):
            if bits == 2 and amb:
                continue
            n = 5000
            d = ctx.alloc(n * 8)
            rc = ctx.lib.kmers_synth_dna(ctx.handle, 0xABCDEF, 12345, n, bits, amb, d)
            assert rc == 0
            got = np.zeros(n, dtype=np.uint64)
            ctx.d2h(got, d)
            ctx.free(d)
            assert np.array_equal(got, orc.synth_words(0xABCDEF, 12345, n, bits, amb))


def test_bad_arguments(km, ctx):
    cap = km._capi
    words = naive.longseq_words("ACGTACGT", 4)
    seq, keep = make_seq(km, words, 8, 4)
    out = np.zeros((8, 1), dtype=np.uint64)
    res = cap.Result()
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 0, 2, vp(out), None, 0, C.byref(res)) == cap.E_BADARG
    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), 3, 0, 2, vp(out), 0, C.byref(res)) == cap.E_BADARG
    # kmers longer than the sequence: an empty iteration whatever their width (FwKmers.jl:63)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 129, 2, vp(out), None, 0, C.byref(res)) == 0 and res.n_out == 0
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 65, 4, vp(out), None, 0, C.byref(res)) == 0 and res.n_out == 0
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 3, 8, vp(out), None, 0, C.byref(res)) == cap.E_BADARG   # no 8-bit kmer alphabet here
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 3, 2, vp(out), None, cap.ASYNC, C.byref(res)) == cap.E_BADARG


def test_minhash_sketch(km, ctx, orc):
    """Fused bottom-s MinHash of fx_hash(canonical kmer) (docs/src/minhash.md:31-35) == the s smallest
    distinct hashes of the materialised iteration."""
    cap = km._capi
    rng = np.random.default_rng(5)
    for bits in (2, 4):
        for K in (5, 16, 31, 33):
            for L, s in ((K - 1, 10), (K, 10), (2000, 50), (2000, 5000), (300_000, 1000), (3_000_000, 1000), (3_000_000, 20_000)):
                if L < 0:
                    continue
                words = orc.synth_words(3 + K, 0, (L * bits + 63) // 64 + 1, bits)
                seq, keep = make_seq(km, words, L, bits)
                out = np.zeros(s, dtype=np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 7, s, vp(out), cap.MEM_HOST, C.byref(res))
                assert rc == 0, ctx.last_error()
                _, eh, _ = orc.canonical(words, L, bits, 2, K, seed=7)
                exp = np.unique(eh)[:s]
                assert res.n_out == len(exp) and np.array_equal(out[:len(exp)], exp), (bits, K, L, s)
    # the host-feedback path (used for s > 4096, and as the overflow fallback) gives the same sketch
    L, K = 1_000_000, 21
    words = orc.synth_words(77, 0, (L * 4 + 63) // 64 + 1, 4)
    seq, keep = make_seq(km, words, L, 4)
    _, eh, _ = orc.canonical(words, L, 4, 2, K, seed=1)
    for s in (1, 2, 3, 100, 4096):
        outs = []
        for host_only in (0, 1):
            ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, host_only)
            out = np.zeros(s, dtype=np.uint64)
            res = cap.Result()
            assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 1, s, vp(out), cap.MEM_HOST, C.byref(res)) == 0
            outs.append(out[:res.n_out].copy())
        ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, 0)
        exp = np.unique(eh)[:s]
        assert np.array_equal(outs[0], exp) and np.array_equal(outs[1], exp), s
    # adversarial order for the device path: hashes that keep decreasing overflow its buffer -> fallback
    text = "".join("ACGT"[(i >> (2 * j)) & 3] for i in range(70000, 0, -1) for j in range(7, -1, -1))
    words = naive.longseq_words(text, 2)
    seq, keep = make_seq(km, words, len(text), 2)
    out = np.zeros(1000, dtype=np.uint64)
    res = cap.Result()
    assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 8, 2, 0, 1000, vp(out), cap.MEM_HOST, C.byref(res)) == 0
    _, eh, _ = orc.canonical(words, len(text), 2, 2, 8)
    exp = np.unique(eh)[:1000]
    assert res.n_out == len(exp) and np.array_equal(out[:len(exp)], exp)
    # low-complexity input: few distinct kmers, many duplicates
    text = "ACGTTGCA" * 50_000
    words = naive.longseq_words(text, 4)
    seq, keep = make_seq(km, words, len(text), 4)
    out = np.zeros(1000, dtype=np.uint64)
    res = cap.Result()
    assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 21, 2, 0, 1000, vp(out), cap.MEM_HOST, C.byref(res)) == 0
    _, eh, _ = orc.canonical(words, len(text), 4, 2, 21)
    exp = np.unique(eh)
    assert res.n_out == len(exp) <= 8 and np.array_equal(out[:len(exp)], exp)
    # genome-like mix: random sequence interrupted by poly-A / poly-T runs and short tandem repeats, whose
    # few small hashes (poly-A hashes to 0) recur throughout; they are dropped against the running sketch
    rng = np.random.default_rng(11)
    parts = []
    for _ in range(300):
        parts.append(naive.random_text(rng, int(rng.integers(2000, 12000))))
        parts.append(str(rng.choice(["A", "T", "AC", "AAG", "ACGT"])) * int(rng.integers(200, 3000)))
    text = "".join(parts)
    for bits in (2, 4):
        words = naive.longseq_words(text, bits)
        seq, keep = make_seq(km, words, len(text), bits)
        for K, s in ((16, 1000), (31, 200)):
            out = np.zeros(s, dtype=np.uint64)
            res = cap.Result()
            assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, s, vp(out), cap.MEM_HOST, C.byref(res)) == 0
            _, eh, _ = orc.canonical(words, len(text), bits, 2, K)
            exp = np.unique(eh)[:s]
            assert res.n_out == len(exp) and np.array_equal(out[:len(exp)], exp), (bits, K, s)
    # sorted (decreasing hash order would overflow a naive buffer) and an ambiguous symbol far into the sequence
    L = 2_000_000
    words = orc.synth_words(1, 0, (L * 4 + 63) // 64 + 1, 4).copy()
    pos = 1_234_567
    words[(pos * 4) >> 6] |= np.uint64(0xF) << np.uint64((pos * 4) & 63)
    seq, keep = make_seq(km, words, L, 4)
    rc = ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), 31, 2, 0, 1000, vp(out), cap.MEM_HOST, C.byref(res))
    assert rc == cap.E_ENCODE and res.err_pos == pos + 1 and res.err_enc == 0xF


def test_composition(km, ctx, orc):
    """counts[as_integer(kmer)] over FwDNAMers{K} (docs/src/composition.md:28-39)."""
    cap = km._capi
    for bits in (2, 4):
        for K in (1, 4, 6, 8, 11):
            L = 500_000
            words = orc.synth_words(K, 0, (L * bits + 63) // 64 + 1, bits)
            seq, keep = make_seq(km, words, L, bits)
            counts = np.zeros(4 ** K, dtype=np.uint32)
            res = cap.Result()
            assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, vp(counts), cap.MEM_HOST, C.byref(res)) == 0
            fw, _ = orc.fw_kmers(words, L, bits, 2, K)
            exp = np.bincount(fw[:, 0].astype(np.int64), minlength=4 ** K).astype(np.uint32)
            assert np.array_equal(counts, exp), (bits, K)
            assert counts.sum() == L - K + 1
    # K = 12 (global counters), an ASCII source, and an EncodeError through the composition kernel
    L = 300_000
    words = orc.synth_words(12, 0, L // 16 + 1, 4)
    seq, keep = make_seq(km, words, L, 4)
    counts = np.zeros(4 ** 12, dtype=np.uint32)
    res = cap.Result()
    assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 12, vp(counts), cap.MEM_HOST, C.byref(res)) == 0
    fw, _ = orc.fw_kmers(words, L, 4, 2, 12)
    assert np.array_equal(counts, np.bincount(fw[:, 0].astype(np.int64), minlength=4 ** 12).astype(np.uint32))
    # K = 13 into a host array; K = 16 (2^32 counters, 16 GiB, the widest index the kernel forms) resident in HBM, read back in pieces
    L = 2_000_000
    words = orc.synth_words(1316, 0, L // 16 + 1, 4)
    seq, keep = make_seq(km, words, L, 4)
    counts = np.zeros(4 ** 13, dtype=np.uint32)
    assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 13, vp(counts), cap.MEM_HOST, C.byref(res)) == 0
    fw, _ = orc.fw_kmers(words, L, 4, 2, 13)
    assert np.array_equal(counts, np.bincount(fw[:, 0].astype(np.int64), minlength=4 ** 13).astype(np.uint32))
    d_w, d_c = ctx.alloc(words.nbytes), ctx.alloc(4 ** 16 * 4)
    ctx.h2d(d_w, words)
    dseq = cap.Seq(d_w, L, 0, 0, 4, 0)
    assert ctx.lib.kmers_composition(ctx.handle, C.byref(dseq), 16, d_c, cap.MEM_DEVICE, C.byref(res)) == 0, ctx.last_error()
    fw, _ = orc.fw_kmers(words, L, 4, 2, 16)
    want_idx, want_cnt = np.unique(fw[:, 0], return_counts=True)
    piece = np.zeros(1 << 26, dtype=np.uint32)          # 256 MiB at a time
    got_idx, got_cnt = [], []
    for i in range(4 ** 16 // len(piece)):
        ctx.d2h(piece, d_c + i * piece.nbytes)
        nz = np.flatnonzero(piece)
        got_idx.append(nz.astype(np.uint64) + np.uint64(i * len(piece)))
        got_cnt.append(piece[nz].astype(np.int64))
    ctx.free(d_w)
    ctx.free(d_c)
    assert np.array_equal(np.concatenate(got_idx), want_idx) and np.array_equal(np.concatenate(got_cnt), want_cnt)
    assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 17, vp(counts), cap.MEM_HOST, C.byref(res)) == cap.E_UNSUPPORTED
    text = naive.random_text(np.random.default_rng(3), 70_001)
    aw = naive.ascii_words(text)
    seq = cap.Seq(aw.ctypes.data, len(text), 0, 0, 8, 0)
    counts = np.zeros(4 ** 8, dtype=np.uint32)
    assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 8, vp(counts), cap.MEM_HOST, C.byref(res)) == 0
    exp = np.bincount(np.array([w[0] for w in naive.fw_kmers(text, 8, 2)], dtype=np.int64), minlength=4 ** 8)
    assert np.array_equal(counts, exp.astype(np.uint32))
    bad = text[:40_000] + "N" + text[40_001:]
    w4 = naive.longseq_words(bad, 4)
    seq, keep = make_seq(km, w4, len(bad), 4)
    rc = ctx.lib.kmers_composition(ctx.handle, C.byref(seq), 8, vp(counts), cap.MEM_HOST, C.byref(res))
    assert rc == cap.E_ENCODE and res.err_pos == 40_001 and res.err_enc == 0xF


def test_composition_counter_overflow_paths(km, ctx, orc):
    """The LDS histograms hold 16-bit counters: low-complexity input (every kmer in one bin, or half of
    them) must take the measure-and-flush path and still count exactly.  KMERS_PARAM_MAX_GRID = 2 makes
    two workgroups walk all the tiles, so 1.2 M kmers are enough to overflow a counter many times."""
    cap = km._capi
    ctx.set_param(cap.PARAM_MAX_GRID, 2)
    try:
        L = 1_200_000
        rng = np.random.default_rng(9)
        half = naive.random_text(rng, L // 2)
        for name, text in (("polyA", "A" * L), ("polyT", "T" * L), ("half", "A" * (L // 2) + half),
                           ("period2", "AC" * (L // 2))):
            for bits in (2, 4):
                words = naive.longseq_words(text, bits)
                for K in (1, 3, 8, 9):
                    seq, keep = make_seq(km, words, L, bits)
                    counts = np.zeros(4 ** K, dtype=np.uint32)
                    res = cap.Result()
                    assert ctx.lib.kmers_composition(ctx.handle, C.byref(seq), K, vp(counts), cap.MEM_HOST, C.byref(res)) == 0
                    fw, _ = orc.fw_kmers(words, L, bits, 2, K)
                    exp = np.bincount(fw[:, 0].astype(np.int64), minlength=4 ** K).astype(np.uint32)
                    bad = np.nonzero(counts != exp)[0]
                    assert len(bad) == 0, (name, bits, K, bad[:6].tolist(), counts[bad[:6]].tolist(), exp[bad[:6]].tolist())
    finally:
        ctx.set_param(cap.PARAM_MAX_GRID, 0)


def test_tuple_layouts(km, ctx, orc):
    """KMERS_OUT_TUPLES: outputs laid out as the eltype of the tuple-yielding iterators --
    Vector{Tuple{Kmer,Kmer}} (FwRvIterator, CanonicalKmers.jl:44-45), Vector{Tuple{Kmer,UInt64}},
    Vector{Tuple{Kmer,Int}} (UnambiguousKmers.jl:39-41) -- equal the interleaved separate arrays."""
    cap = km._capi
    rng = np.random.default_rng(123)
    for bits in (2, 4):
        for K in (5, 31, 33, 64, 70):
            N = (2 * K + 63) // 64
            for L in (K, 1000, 9973):
                words = orc.synth_words(K + bits, 0, (L * bits + 63) // 64 + 1, bits)
                n = L - K + 1
                seq, keep = make_seq(km, words, L, bits)
                res = cap.Result()
                efw, erv, _ = orc.fwrv(words, L, bits, 2, K)
                ek, eh, _ = orc.canonical(words, L, bits, 2, K, seed=11)
                t = np.zeros((n, 2 * N), dtype=np.uint64)
                rc = ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, vp(t), None, cap.OUT_TUPLES, C.byref(res))
                assert rc == 0, ctx.last_error()
                assert np.array_equal(t, np.concatenate([efw, erv], axis=1)), (bits, K, L)
                t = np.zeros((n, N + 1), dtype=np.uint64)
                rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, vp(t), None, 11, cap.OUT_TUPLES, C.byref(res))
                assert rc == 0, ctx.last_error()
                assert np.array_equal(t, np.concatenate([ek, eh[:, None]], axis=1)), (bits, K, L)
    for K in (4, 31, 40):
        N = (2 * K + 63) // 64
        text = naive.random_text(rng, 20000, p_amb=0.03)
        words = naive.longseq_words(text, 4)
        seq, keep = make_seq(km, words, len(text), 4)
        res = cap.Result()
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, None, None, 0, 0, C.byref(res)) == 0
        m = int(res.n_out)
        t = np.zeros((m, N + 1), dtype=np.uint64)
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, vp(t), None, m, cap.OUT_TUPLES, C.byref(res)) == 0
        ek, es, _ = orc.unambiguous(words, len(text), 4, K)
        assert np.array_equal(t, np.concatenate([ek, es.astype(np.uint64)[:, None]], axis=1))
    # misuse
    out = np.zeros((10, 2), dtype=np.uint64)
    seq, keep = make_seq(km, naive.longseq_words("ACGTACGTAC", 4), 10, 4)
    res = cap.Result()
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), 3, 2, vp(out), vp(out), cap.OUT_TUPLES, C.byref(res)) == cap.E_BADARG
    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), 3, 2, 2, vp(out), cap.OUT_TUPLES, C.byref(res)) == cap.E_BADARG


def test_minimizers(km, ctx, orc):
    """kmers_minimizers vs the oracle, both modes (docs/src/replacements.md:33-51, test/benchmark.jl:96-119)."""
    cap = km._capi
    for src in (2, 4):
        for K, W, stride in [(8, 20, 20), (5, 9, 1), (31, 10, 7), (33, 5, 3), (64, 3, 40), (21, 50, 11), (4, 1, 2)]:
            N = (2 * K + 63) // 64
            for L in (K + W - 2, K + W - 1, 5000, 100_003):
                words = orc.synth_words(K * W, 0, (L * src + 63) // 64 + 1, src)
                span = K + W - 1
                n = 0 if L < span else (L - span) // stride + 1
                for mode in (0, 1):
                    out = np.zeros((max(n, 1), N), dtype=np.uint64)
                    res = cap.Result()
                    seq, keep = make_seq(km, words, L, src)
                    rc = ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), K, W, stride, 2, mode, vp(out), cap.MEM_HOST, C.byref(res))
                    assert rc == 0, (K, W, stride, L, ctx.last_error())
                    exp, eres = orc.minimizers(words, L, src, 2, K, W, stride, mode)
                    assert res.n_out == n == len(exp)
                    assert np.array_equal(out[:n], exp), (src, K, W, stride, L, mode)
    # the benchmark's shape: K = 8, W = 20, windows every 20 symbols of a 2-bit sequence, XOR of data[1]
    L = 10_000_000
    words = orc.synth_words(439824, 0, L * 2 // 64 + 1, 2)
    n = (L - 27) // 20 + 1
    out = np.zeros((n, 1), dtype=np.uint64)
    res = cap.Result()
    seq, keep = make_seq(km, words, L, 2)
    assert ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), 8, 20, 20, 2, 0, vp(out), cap.MEM_HOST, C.byref(res)) == 0
    exp, _ = orc.minimizers(words, L, 2, 2, 8, 20, 20, 0)
    assert int(np.bitwise_xor.reduce(out[:, 0])) == int(np.bitwise_xor.reduce(exp[:, 0]))
    # an ambiguous symbol inside a window is an EncodeError at its position
    text = "ACGT" * 30 + "N" + "ACGT" * 30
    words = naive.longseq_words(text, 4)
    seq, keep = make_seq(km, words, len(text), 4)
    rc = ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), 5, 4, 3, 2, 1, vp(out), cap.MEM_HOST, C.byref(res))
    assert rc == cap.E_ENCODE and res.err_pos == 121 and res.err_enc == 0xF
    # K <= stride < K + W - 1: every symbol is still read by some window; stride >= K + W - 1: gaps are not
    rng = np.random.default_rng(4)
    for K, W, stride in [(8, 20, 20), (8, 20, 26), (8, 20, 27), (8, 20, 40), (3, 2, 5)]:
        for _ in range(6):
            L = 3000
            t = list(naive.random_text(rng, L))
            for p in rng.integers(0, L, size=2):
                t[p] = "N"
            words = naive.longseq_words("".join(t), 4)
            seq, keep = make_seq(km, words, L, 4)
            n = (L - (K + W - 1)) // stride + 1
            o = np.zeros((n, 1), dtype=np.uint64)
            rc = ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), K, W, stride, 2, 0, vp(o), cap.MEM_HOST, C.byref(res))
            exp, eres = orc.minimizers(words, L, 4, 2, K, W, stride, 0)
            if eres.status == 0:
                assert rc == 0 and np.array_equal(o, exp)
            else:
                assert rc == cap.E_ENCODE and (res.err_pos, res.err_enc) == (eres.err_pos, eres.err_enc), (K, W, stride)


def test_async_device_calls_and_sync(km, ctx, orc):
    """KMERS_MEM_DEVICE | KMERS_ASYNC enqueues and returns; kmers_sync() reports the first EncodeError
    seen since the last sync (position + encoding), then the context is clean again."""
    cap = km._capi
    L, K = 300_000, 31
    words = orc.synth_words(99, 0, (L * 4 + 63) // 64 + 1, 4).copy()
    n = L - K + 1
    dw, dk, dh = ctx.alloc(words.nbytes + 8), ctx.alloc(n * 8), ctx.alloc(n * 8)
    ctx.h2d(dw, words)
    seq = cap.Seq(dw, L, 0, 0, 4, 0)
    res = cap.Result()
    flags = cap.MEM_DEVICE | cap.ASYNC
    for _ in range(3):
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, dk, dh, 0, flags, C.byref(res)) == 0
    rc, sres = ctx.sync()
    assert rc == 0 and sres.status == 0
    got = np.zeros(n, np.uint64)
    ctx.d2h(got, dk)
    ek, _, _ = orc.canonical(words, L, 4, 2, K)
    assert np.array_equal(got, ek[:, 0])
    # poison symbol 200001 (1-based) with a gap, run async, collect at sync
    pos = 200_000
    words[(pos * 4) >> 6] &= ~(np.uint64(0xF) << np.uint64((pos * 4) & 63))
    ctx.h2d(dw, words)
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, dk, dh, 0, flags, C.byref(res)) == 0
    rc, sres = ctx.sync()
    assert rc == cap.E_ENCODE and sres.err_pos == pos + 1 and sres.err_enc == 0
    rc, sres = ctx.sync()
    assert rc == 0
    for p in (dw, dk, dh):
        ctx.free(p)


def test_reduce_xor_over_spaced_and_unambiguous(km, ctx, orc):
    """kmers_reduce_xor_iter: the XOR reducer of test/benchmark.jl:9-15 fused over SpacedKmers and
    UnambiguousKmers == XOR over the head words of the materialised iteration."""
    cap = km._capi
    rng = np.random.default_rng(77)
    for src in (2, 4, 8):
        for L in (0, 5, 40, 1000, 70_001):
            text = naive.random_text(rng, L, p_amb=0.01 if src != 2 else 0.0)
            clean = naive.random_text(rng, L)
            for K in (1, 7, 21, 32, 33, 64):
                for name, t in (("amb", text), ("clean", clean)):
                    words = naive.ascii_words(t) if src == 8 else naive.longseq_words(t if t else "A", src)
                    seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                    res = cap.Result()
                    val = C.c_uint64(123)
                    rc = ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_UNAMBIGUOUS, 1, C.byref(val), 0, C.byref(res))
                    ek, _, eres = orc.unambiguous(words, L, src, K)
                    assert rc == 0, (src, L, K, name, ctx.last_error())
                    exp = int(np.bitwise_xor.reduce(ek[:, 0])) if len(ek) else 0
                    assert val.value == exp, (src, L, K, name)
                    for J in (1, 5, 7, 32):
                        ek, eres = orc.spaced(words, L, src, 2, K, J)
                        rc = ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_SPACED, J, C.byref(val), 0, C.byref(res))
                        if eres.status == 1:
                            assert rc == cap.E_ENCODE and res.err_pos == eres.err_pos and res.err_enc == eres.err_enc, (src, L, K, J, name)
                        else:
                            assert rc == 0 and val.value == (int(np.bitwise_xor.reduce(ek[:, 0])) if len(ek) else 0), (src, L, K, J, name)
    # 4-bit kmers through the spaced reducer; fw / canonical are forwarded to kmers_reduce_xor
    t = naive.random_text(rng, 5000, p_amb=0.05)
    words = naive.longseq_words(t, 4)
    seq = cap.Seq(words.ctypes.data, len(t), 0, 0, 4, 0)
    res, val = cap.Result(), C.c_uint64()
    ek, _ = orc.spaced(words, len(t), 4, 4, 9, 4)
    assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), 9, 4, cap.ITER_SPACED, 4, C.byref(val), 0, C.byref(res)) == 0
    assert val.value == int(np.bitwise_xor.reduce(ek[:, 0]))
    exp, _ = orc.reduce_xor_canonical(words, len(t), 4, 4, 9)
    assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), 9, 4, cap.ITER_CANONICAL, 1, C.byref(val), 0, C.byref(res)) == 0
    assert val.value == exp
    # strides no tile can stage (J * bits > 64) run one lane per kmer (tests/test_gpu_wide.py); an ambiguous symbol inside a
    # window is still the iterator's EncodeError
    rc = ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), 9, 2, cap.ITER_SPACED, 40, C.byref(val), 0, C.byref(res))
    _, eres = orc.spaced(words, len(t), 4, 2, 9, 40)
    assert rc == cap.E_ENCODE and (res.err_pos, res.err_enc) == (eres.err_pos, eres.err_enc)


def test_unambiguous_lookback_gives_up_instead_of_hanging(km):
    """The give-up path of the one-pass kernel's look-back (unambiguous_kernel.hpp: SPIN_LIMIT, the abort flag) cannot be provoked
    on a healthy device, so a TEST BUILD of the same library provokes it: libkmers_hip_testabort.so (-DKMERS_TEST_ABORT,
    kmers_jl_amd/build.py) never publishes the count of tile 1.  The launch must come back -- every look-back sees the flag and
    drains -- with KMERS_E_HIP and a message, and the device must be usable afterwards.  A fresh process: one library per process."""
    import os
    import subprocess
    import sys
    from kmers_jl_amd import build
    lib = build.TEST_ABORT_LIB
    assert os.path.exists(lib), "run __graft_entry__.build() first: it builds the test library next to the product"
    code = r'''
import ctypes as C, sys
import numpy as np
import kmers_jl_amd as km
cap = km._capi
ctx = km.Context(0)
L, K = 400_000, 31                      # nine tiles of 49152 candidate starts (any tile length below L / 2 leaves a tile 1 AND a tile 2 to wait for it)
nw = L // 16 + 1
d_w, d_k, d_s = ctx.alloc(nw * 8 + 8), ctx.alloc(L * 8), ctx.alloc(L * 8)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 3, 0, nw, 4, 2621, d_w), "synth")
res = cap.Result()
seq = cap.Seq(d_w, L, 0, 0, 4, 0)
rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, d_k, d_s, L, cap.MEM_DEVICE, C.byref(res))
assert rc == cap.E_HIP, rc
assert "gave up" in ctx.last_error(), ctx.last_error()
# one tile: nothing to look back at -- the same build, the same context, right behind the aborted launch
seq1 = cap.Seq(d_w, 20_000, 0, 0, 4, 0)
rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq1), K, 1, d_k, d_s, L, cap.MEM_DEVICE, C.byref(res))
assert rc == 0 and res.n_out > 0, (rc, ctx.last_error())
print("ok", res.n_out)
'''
    env = dict(os.environ, KMERS_HIP_LIB=lib, PYTHONPATH=os.pathsep.join(sys.path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr[-2000:])
