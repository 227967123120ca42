"""Kmers of more than four words through EVERY entry point.  Kmer{A,K,N} has no upper bound on N (src/kmer.jl:97-111:
N = cld(K * bits_per_symbol, 64)); the tile kernels are compiled for N = 1..4, everything wider runs on run-time-width
kernels (wide_kernel.hpp, transform_kernel_any, ragged_wide_kernel, record_sketch_kernel<.., 0, ..>).  Bit-exact against the
oracle (up to its 64 words), and against the independent big-integer slicer (tests/naive.py)."""
import ctypes as C

import numpy as np
import pytest

import naive

pytestmark = pytest.mark.gpu

ONES = np.uint64(0xFFFFFFFFFFFFFFFF)


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    return kmers_jl_amd


@pytest.fixture(scope="module", params=[0, 1], ids=["tile-form", "lane-per-kmer"])
def ctx(km, request):
    """Both kernels behind every wide launch at stride 1: the tile form (wide_tile_kernel.hpp, the default) and the
    one-lane-per-kmer kernel it falls back to (wide_kernel.hpp; KMERS_PARAM_WIDE_NO_TILES)."""
    c = km.Context(0)
    c.set_param(km._capi.PARAM_WIDE_NO_TILES, request.param)
    yield c
    c.close()


def vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def source_words(text, src):
    return naive.ascii_words(text) if src == 8 else naive.longseq_words(text if text else "A", src)


# (source bits, kmer alphabet bits, K): every RecodingScheme, five to eight words and -- the oracle goes to 64 words, checked
# against the naive slicer in tests/test_oracle_property.py -- 19, 32 and 63
WIDE = [(2, 2, 129), (2, 2, 256), (4, 2, 160), (4, 2, 255), (4, 4, 65), (4, 4, 128), (2, 4, 100), (8, 2, 130), (8, 4, 70),
        (4, 2, 600), (2, 4, 512), (8, 2, 2001)]


def test_transforms_of_wide_kmers(km, ctx, orc):
    """reverse / complement / reverse_complement / canonical / iscanonical / LongSequence(kmer) / count(isGC) on arrays of
    kmers of five to eight words vs the oracle (transformations.jl:1-41, construction.jl:289-324, counting.jl:1-8); 22 words
    vs text operations; overlapping device arrays."""
    cap = km._capi
    rng = np.random.default_rng(8)
    for bits, ks in ((2, (129, 160, 200, 255, 256, 1000, 2048)), (4, (65, 80, 100, 127, 128, 333, 1024))):
        for K in ks:
            N = (K * bits + 63) // 64
            assert 4 < N <= 64
            n = 257
            texts = [naive.random_text(rng, K, p_amb=0.2 if bits == 4 else 0.0) for _ in range(n)]
            texts[0] = ("ACGT" * K)[:K // 2] + naive.revcomp_text(("ACGT" * K)[:K // 2]) + ("" if K % 2 == 0 else "A")  # (nearly) its own reverse complement
            arr = np.array([naive.kmer_words(t, bits) for t in texts], dtype=np.uint64).reshape(n, N)
            for op, fn in ((cap.OP_REVERSE, orc.reverse), (cap.OP_COMPLEMENT, orc.complement),
                           (cap.OP_REVCOMP, orc.reverse_complement), (cap.OP_CANONICAL, orc.canonical_kmer)):
                out = np.zeros((n, N), dtype=np.uint64)
                assert ctx.lib.kmers_transform(ctx.handle, op, vp(arr), K, bits, n, vp(out), cap.MEM_HOST) == 0, ctx.last_error()
                assert [tuple(r) for r in out.tolist()] == [fn(tuple(r), K, bits) for r in arr.tolist()], (bits, K, op)
            flags = np.zeros(n, dtype=np.uint64)
            assert ctx.lib.kmers_transform(ctx.handle, cap.OP_ISCANONICAL, vp(arr), K, bits, n, vp(flags), cap.MEM_HOST) == 0
            assert flags.astype(bool).tolist() == [orc.iscanonical(tuple(r), K, bits) for r in arr.tolist()]
            ls = np.zeros((n, N), dtype=np.uint64)
            assert ctx.lib.kmers_transform(ctx.handle, cap.OP_TO_LONGSEQ, vp(arr), K, bits, n, vp(ls), cap.MEM_HOST) == 0
            assert [list(r) for r in ls.tolist()] == [list(naive.longseq_words(t, bits)[:N]) for t in texts], (bits, K)
            if bits == 2:
                gc = np.zeros(n, dtype=np.uint64)
                assert ctx.lib.kmers_transform(ctx.handle, cap.OP_COUNT_GC, vp(arr), K, bits, n, vp(gc), cap.MEM_HOST) == 0
                assert gc.tolist() == [sum(c in "GC" for c in t) for t in texts]
            # as_integer / from_integer stay refused above 128 bits, with the reference's message (kmer.jl:324)
            assert ctx.lib.kmers_transform(ctx.handle, cap.OP_AS_INTEGER, vp(arr), K, bits, n, vp(ls), cap.MEM_HOST) == cap.E_BADARG
    # 22 words: the oracle stops at 8, text operations do not
    K, n = 700, 40
    texts = [naive.random_text(rng, K) for _ in range(n)]
    N = (2 * K + 63) // 64
    arr = np.array([naive.kmer_words(t, 2) for t in texts], dtype=np.uint64).reshape(n, N)
    out = np.zeros((n, N), dtype=np.uint64)
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, vp(arr), K, 2, n, vp(out), cap.MEM_HOST) == 0
    assert [tuple(r) for r in out.tolist()] == [tuple(naive.kmer_words(naive.revcomp_text(t), 2)) for t in texts]
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_REVERSE, vp(arr), K, 2, n, vp(out), cap.MEM_HOST) == 0
    assert [tuple(r) for r in out.tolist()] == [tuple(naive.kmer_words(t[::-1], 2)) for t in texts]
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_CANONICAL, vp(arr), K, 2, n, vp(out), cap.MEM_HOST) == 0
    want = [min(tuple(naive.kmer_words(t, 2)), tuple(naive.kmer_words(naive.revcomp_text(t), 2))) for t in texts]
    assert [tuple(r) for r in out.tolist()] == want
    # device arrays that overlap (in place): a result word depends on two input words, the library goes through a copy
    d = ctx.alloc(arr.nbytes)
    ctx.h2d(d, arr)
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, d, K, 2, n, d, cap.MEM_DEVICE) == 0
    ctx.d2h(out, d)
    ctx.free(d)
    assert [tuple(r) for r in out.tolist()] == [tuple(naive.kmer_words(naive.revcomp_text(t), 2)) for t in texts]


def test_tuple_layouts_of_wide_kmers(km, ctx, orc):
    """KMERS_OUT_TUPLES with kmers of more than four words: Tuple{Kmer,Kmer} and Tuple{Kmer,UInt64} elements
    (CanonicalKmers.jl:44-45)."""
    cap = km._capi
    rng = np.random.default_rng(9)
    for src, dst, K in WIDE:
        N = (K * dst + 63) // 64
        for L in (K, K + 1, max(1777, K + 300)):
            text = naive.random_text(rng, L, p_amb=0.05 if dst == 4 and src != 2 else 0.0)
            words = source_words(text, src)
            seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
            n = L - K + 1
            res = cap.Result()
            efw, erv, _ = orc.fwrv(words, L, src, dst, K)
            ek, eh, _ = orc.canonical(words, L, src, dst, K, seed=5)
            t = np.zeros((n, 2 * N), dtype=np.uint64)
            assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(t), None, cap.OUT_TUPLES, C.byref(res)) == 0, ctx.last_error()
            assert np.array_equal(t, np.concatenate([efw, erv], axis=1)), (src, dst, K, L)
            t = np.zeros((n, N + 1), dtype=np.uint64)
            assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(t), None, 5, cap.OUT_TUPLES, C.byref(res)) == 0
            assert np.array_equal(t, np.concatenate([ek, eh[:, None]], axis=1)), (src, dst, K, L)


def test_fused_reducers_over_wide_kmers(km, ctx, orc):
    """kmers_reduce_xor / kmers_reduce_xor_iter (test/benchmark.jl:9-15) over FwKmers, CanonicalKmers, SpacedKmers and
    UnambiguousKmers of more than four words == XOR of data[1] of the materialised iteration; and SpacedKmers of ANY width
    at strides no tile can stage (J * bits > 64)."""
    cap = km._capi
    rng = np.random.default_rng(10)
    for src, dst, K in WIDE:
        for L in (K - 1, K, K + 70, 6000):
            text = naive.random_text(rng, L, p_amb=0.05 if dst == 4 and src != 2 else 0.0)
            words = source_words(text, src)
            seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
            res, val = cap.Result(), C.c_uint64(99)
            efw, _ = orc.fw_kmers(words, L, src, dst, K)
            ek, _, _ = orc.canonical(words, L, src, dst, K)
            fold = lambda a: int(np.bitwise_xor.reduce(a[:, 0])) if len(a) else 0
            assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, dst, 0, C.byref(val), 0, C.byref(res)) == 0, ctx.last_error()
            assert val.value == fold(efw), (src, dst, K, L)
            assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, dst, 1, C.byref(val), 0, C.byref(res)) == 0
            assert val.value == fold(ek), (src, dst, K, L)
            for J in (1, 3, 40, K + 9):
                es, _ = orc.spaced(words, L, src, dst, K, J)
                assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, dst, cap.ITER_SPACED, J, C.byref(val), 0, C.byref(res)) == 0
                assert val.value == fold(es) and res.n_out == len(es), (src, dst, K, L, J)
    # UnambiguousKmers (2-bit kmers; skips instead of failing)
    for src, K in ((4, 129), (4, 200), (8, 150), (2, 256)):
        for L in (K - 1, K, 5000):
            text = naive.random_text(rng, L, p_amb=0.002 if src != 2 else 0.0)
            words = source_words(text, src)
            seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
            res, val = cap.Result(), C.c_uint64(99)
            uk, _, _ = orc.unambiguous(words, L, src, K)
            assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_UNAMBIGUOUS, 1, C.byref(val), 0, C.byref(res)) == 0
            assert val.value == (int(np.bitwise_xor.reduce(uk[:, 0])) if len(uk) else 0), (src, K, L)
    # narrow kmers, strides beyond a tile's reach (round 2 refused these)
    for src, dst, K, J in ((4, 2, 9, 40), (4, 2, 31, 1000), (2, 4, 5, 17), (8, 2, 21, 33), (4, 4, 40, 64)):
        L = 20_011
        text = naive.random_text(rng, L)
        words = source_words(text, src)
        seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
        res, val = cap.Result(), C.c_uint64()
        es, _ = orc.spaced(words, L, src, dst, K, J)
        assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, dst, cap.ITER_SPACED, J, C.byref(val), 0, C.byref(res)) == 0
        assert val.value == int(np.bitwise_xor.reduce(es[:, 0])), (src, dst, K, J)
        # ... and materialised: kmers_spaced (the tile form of the run-time-width kernel, or the symbol-by-symbol gather kernel)
        sp = np.zeros(es.shape, np.uint64)
        assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(sp), cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
        assert res.n_out == len(es) and np.array_equal(sp, es), (src, dst, K, J)
    # strictness is the iterator's: the first symbol the 2-bit alphabet cannot hold, in sequence order; with J >= K a symbol
    # between two windows is never inspected (SpacedKmers.jl:133-134)
    K, L = 130, 3000
    t = list(naive.random_text(rng, L))
    t[2000], t[700] = "N", "W"
    words = naive.longseq_words("".join(t), 4)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    res, val = cap.Result(), C.c_uint64()
    for canonical in (0, 1):
        rc = ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, canonical, C.byref(val), 0, C.byref(res))
        assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, 701, 0b1001)
    t = list(naive.random_text(rng, L))
    t[135] = "N"                                   # K = 130, J = 140: symbols 131..140 lie between the first two windows
    words = naive.longseq_words("".join(t), 4)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    es, eres = orc.spaced(words, L, 4, 2, K, 140)
    assert eres.status == 0
    assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, 2, cap.ITER_SPACED, 140, C.byref(val), 0, C.byref(res)) == 0
    assert val.value == int(np.bitwise_xor.reduce(es[:, 0]))


def test_minhash_of_wide_kmers(km, ctx, orc):
    """kmers_minhash over canonical kmers of more than four words == the s smallest distinct fx_hash values of the
    materialised iteration (docs/src/minhash.md:17-41), both the device-resident and the host-feedback path."""
    cap = km._capi
    rng = np.random.default_rng(11)
    for src, dst, K in WIDE:
        for L, s in ((K - 1, 10), (K, 10), (3000, 50), (3000, 5000), (200_000, 300)):
            text = naive.random_text(rng, L, p_amb=0.05 if dst == 4 and src != 2 else 0.0)
            words = source_words(text, src)
            seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
            _, eh, _ = orc.canonical(words, L, src, dst, K, seed=7)
            exp = np.unique(eh)[:s]
            for host_only in (0, 1):
                ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, host_only)
                out = np.zeros(s, dtype=np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, dst, 7, s, vp(out), cap.MEM_HOST, C.byref(res))
                ctx.set_param(cap.PARAM_SKETCH_HOST_ONLY, 0)
                assert rc == 0, ctx.last_error()
                assert res.n_out == len(exp) and np.array_equal(out[:len(exp)], exp), (src, dst, K, L, s, host_only)
    # an ambiguous symbol is the iterator's EncodeError
    K, L = 140, 50_000
    t = list(naive.random_text(rng, L))
    t[31_000] = "N"
    words = naive.longseq_words("".join(t), 4)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    out = np.zeros(100, dtype=np.uint64)
    res = cap.Result()
    rc = ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 100, vp(out), cap.MEM_HOST, C.byref(res))
    assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, 31_001, 0xF)


def test_minimizers_of_wide_kmers(km, ctx, orc):
    """kmers_minimizers (docs/src/replacements.md:33-51, test/benchmark.jl:96-110), both modes, kmers of more than four
    words; and narrow kmers at window strides beyond a tile's reach."""
    cap = km._capi
    rng = np.random.default_rng(12)
    cases = [(src, dst, K, W, stride) for src, dst, K in WIDE for W, stride in ((1, 1), (7, 3), (20, 200))]
    cases += [(4, 2, 8, 20, 300), (2, 2, 31, 5, 700), (4, 4, 9, 4, 200), (8, 2, 21, 3, 520)]
    for src, dst, K, W, stride in cases:
        N = (K * dst + 63) // 64
        for L in (K + W - 2, K + W - 1, 4001):
            text = naive.random_text(rng, L, p_amb=0.05 if dst == 4 and src != 2 else 0.0)
            words = source_words(text, src)
            span = K + W - 1
            n = 0 if L < span else (L - span) // stride + 1
            for mode in (0, 1):
                out = np.zeros((max(n, 1), N), dtype=np.uint64)
                res = cap.Result()
                seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                rc = ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), K, W, stride, dst, mode, vp(out), cap.MEM_HOST, C.byref(res))
                assert rc == 0, (src, dst, K, W, stride, L, ctx.last_error())
                exp, _ = orc.minimizers(words, L, src, dst, K, W, stride, mode)
                assert res.n_out == n == len(exp)
                assert np.array_equal(out[:n], exp), (src, dst, K, W, stride, L, mode)
    # an ambiguous symbol inside a window is an EncodeError at its position; between two windows it is never read
    K, W, L = 129, 4, 2000
    for stride, pos, fails in ((3, 1500, True), (300, 400, True), (300, 135, False)):   # windows cover [1, 132], [301, 432], ...
        t = list(naive.random_text(rng, L))
        t[pos - 1] = "N"
        words = naive.longseq_words("".join(t), 4)
        seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
        n = (L - (K + W - 1)) // stride + 1
        out = np.zeros((n, 5), dtype=np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), K, W, stride, 2, 1, vp(out), cap.MEM_HOST, C.byref(res))
        exp, eres = orc.minimizers(words, L, 4, 2, K, W, stride, 1)
        assert (eres.status != 0) == fails
        if fails:
            assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, pos, 0xF) == (cap.E_ENCODE, eres.err_pos, eres.err_enc)
        else:
            assert rc == 0 and np.array_equal(out, exp)


def build_pool(km, texts, src, rng, scatter):
    spans, pieces, pos = [], [], 0
    for t in texts:
        gap = naive.random_text(rng, int(rng.integers(0, 40))) if scatter else ""
        pieces.append(gap + t)
        spans.append((pos + len(gap), len(t)))
        pos += len(gap) + len(t)
    whole = "".join(pieces)
    arr = (km._capi.Span * max(len(spans), 1))(*[km._capi.Span(a, b) for a, b in spans])
    return source_words(whole, src), arr, len(whole)


def test_batches_of_wide_kmers(km, ctx, orc):
    """kmers_batch / kmers_batch_spaced / kmers_minhash_batch with kmers of more than four words == the per-record iterators
    of the oracle, concatenated in record order; strict errors name the record and the position; skip mode masks."""
    cap = km._capi
    rng = np.random.default_rng(13)
    for src, dst, K in WIDE:
        N = (K * dst + 63) // 64
        for n_rec, scatter in ((1, False), (40, True), (600, False)):
            lens = rng.choice([0, 1, K - 1, K, K + 1, K + 50, 700, 2500], n_rec)
            texts = [naive.random_text(rng, int(l)) for l in lens]
            words, spans, n_pool = build_pool(km, texts, src, rng, scatter and src != 8)
            seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
            per = [source_words(t, src) for t in texts]
            eoff = np.cumsum([0] + [max(0, len(t) - K + 1) for t in texts]).astype(np.uint64)
            total = int(eoff[-1])
            for mode in (cap.BATCH_FW, cap.BATCH_CANONICAL):
                ea, eb = [], []
                for t, w in zip(texts, per):
                    if len(t) >= K:
                        a, b, r = orc.fwrv(w, len(t), src, dst, K) if mode == cap.BATCH_FW else orc.canonical(w, len(t), src, dst, K, seed=9)
                        ea.append(a)
                        eb.append(b)
                res = cap.Result()
                offs = np.zeros(n_rec + 1, np.uint64)
                out_a = np.zeros((max(total, 1), N), np.uint64)
                out_b = np.zeros((max(total, 1), N) if mode == cap.BATCH_FW else max(total, 1), np.uint64)
                rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, mode, K, dst, vp(out_a), vp(out_b), 9, vp(offs), total, 0,
                                         C.byref(res))
                assert rc == 0 and res.n_out == total and np.array_equal(offs, eoff), (src, dst, K, n_rec, mode, ctx.last_error())
                if total:
                    assert np.array_equal(out_a[:total], np.concatenate(ea)) and np.array_equal(out_b[:total], np.concatenate(eb)), (src, dst, K, n_rec, mode)
            # SpacedKmers of every record
            for J in (5, K + 3):
                exp = [orc.spaced(w, len(t), src, dst, K, J)[0] for t, w in zip(texts, per) if len(t) >= K]
                tot = sum(len(e) for e in exp)
                out = np.zeros((max(tot, 1), N), np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, n_rec, K, J, dst, vp(out), None, tot, 0, C.byref(res))
                assert rc == 0 and res.n_out == tot, ctx.last_error()
                if tot:
                    assert np.array_equal(out[:tot], np.concatenate(exp)), (src, dst, K, n_rec, J)
            # one sketch per record
            s = 64
            sk = np.zeros((n_rec, s), np.uint64)
            cnt = np.zeros(n_rec, np.uint64)
            res = cap.Result()
            rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, n_rec, K, dst, 3, s, vp(sk), vp(cnt), 0, C.byref(res))
            assert rc == 0 and res.n_out == n_rec, ctx.last_error()
            for i, (t, w) in enumerate(zip(texts, per)):
                exp = np.unique(orc.canonical(w, len(t), src, dst, K, seed=3)[1])[:s] if len(t) >= K else np.zeros(0, np.uint64)
                assert cnt[i] == len(exp) and np.array_equal(sk[i, :len(exp)], exp), (src, dst, K, n_rec, i)
    # strict: the first failing record in batch order and its first offending symbol; skip: all-ones elements
    for src in (4, 8):
        K = 150
        texts = [naive.random_text(rng, 400), naive.random_text(rng, 100), naive.random_text(rng, 500), naive.random_text(rng, 300)]
        t2 = list(texts[2])
        t2[320], t2[170] = "N", "W"
        texts[2] = "".join(t2)
        t3 = list(texts[3])
        t3[10] = "N"
        texts[3] = "".join(t3)
        words, spans, n_pool = build_pool(km, texts, src, rng, src != 8)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        N = 5
        total = sum(max(0, len(t) - K + 1) for t in texts)
        out_a, out_b = np.zeros((total, N), np.uint64), np.zeros(total, np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, 4, cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 1, None, total, 0, C.byref(res))
        assert (rc, res.n_out, res.err_pos, res.err_enc) == (cap.E_ENCODE, 2, 171, ord("W") if src == 8 else 0b1001)
        sk, cnt = np.zeros((4, 20), np.uint64), np.zeros(4, np.uint64)
        rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, 4, K, 2, 1, 20, vp(sk), vp(cnt), 0, C.byref(res))
        assert (rc, res.n_out, res.err_pos) == (cap.E_ENCODE, 2, 171)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, 4, cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 1, None, total,
                                 cap.BATCH_SKIP, C.byref(res))
        assert rc == 0 and res.n_out == total
        assert ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, 4, K, 2, 1, 20, vp(sk), vp(cnt), cap.BATCH_SKIP, C.byref(res)) == 0
        lo = 0
        for i, t in enumerate(texts):
            m = max(0, len(t) - K + 1)
            w = source_words(t, src)
            uk, us, _ = orc.unambiguous(w, len(t), src, K)
            kept = np.zeros(m, bool)
            kept[us - 1] = True
            got = out_a[lo:lo + m]
            assert np.array_equal(~(got == ONES).all(axis=1), kept), (src, i)
            canon = np.array([orc.canonical_kmer(tuple(int(x) for x in r), K, 2) for r in uk], dtype=np.uint64).reshape(-1, N)
            assert np.array_equal(got[kept], canon)
            hashes = np.array([orc.fx_hash(tuple(int(x) for x in r), 1) for r in canon], dtype=np.uint64)
            assert np.array_equal(out_b[lo:lo + m][kept], hashes) and (out_b[lo:lo + m][~kept] == ONES).all()
            e = np.unique(hashes)[:20]
            assert cnt[i] == len(e) and np.array_equal(sk[i, :len(e)], e), (src, i)
            lo += m


def test_widths_beyond_the_oracle(km, ctx):
    """22-word kmers (K = 700, 2-bit) through the consumers, against the big-integer slicer of tests/naive.py."""
    cap = km._capi
    rng = np.random.default_rng(14)
    K, L = 700, 1500
    text = naive.random_text(rng, L)
    words = naive.longseq_words(text, 4)
    seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
    canon = naive.canonical(text, K, 2)
    hashes = sorted(set(naive.fx_hash(list(c), 21) for c in canon))
    res, val = cap.Result(), C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(val), 0, C.byref(res)) == 0
    x = 0
    for c in canon:
        x ^= c[0]
    assert val.value == x
    out = np.zeros(100, dtype=np.uint64)
    assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 21, 100, vp(out), cap.MEM_HOST, C.byref(res)) == 0
    assert res.n_out == 100 and out.tolist() == hashes[:100]
    # true sliding-window minimizers: the kmer with the smallest fx_hash among W consecutive forward kmers, leftmost on ties
    W, stride = 6, 50
    fw = naive.fw_kmers(text, K, 2)
    n = (L - (K + W - 1)) // stride + 1
    N = (2 * K + 63) // 64
    mz = np.zeros((n, N), dtype=np.uint64)
    assert ctx.lib.kmers_minimizers(ctx.handle, C.byref(seq), K, W, stride, 2, 1, vp(mz), cap.MEM_HOST, C.byref(res)) == 0
    want = [min(fw[i * stride:i * stride + W], key=lambda kmer: naive.fx_hash(list(kmer), 0)) for i in range(n)]
    assert [tuple(int(v) for v in r) for r in mz] == want


def test_tile_form_at_every_tile_length(km, orc):
    """wide_tile_kernel.hpp with tiles of 1, 7, 64, 333 and 5000 windows (KMERS_PARAM_TILE_KMERS), views that start anywhere
    in a source word, every RecodingScheme: FwKmers + reverse complements, CanonicalKmers + fx_hash, both tuple layouts, the
    XOR reducer and the MinHash sketch against the oracle; the first offending symbol in sequence order whichever tile sees it."""
    cap = km._capi
    ctx = km.Context(0)
    rng = np.random.default_rng(16)
    try:
        for src, dst, K in WIDE:
            N = (K * dst + 63) // 64
            for tile in (1, 7, 64, 333, 5000):
                ctx.set_param(cap.PARAM_TILE_KMERS, tile)
                first = int(rng.choice([0, 1, 5, 15, 16, 31, 32, 33, 63, 64, 100]))
                L = int(rng.choice([K, K + 1, K + 63, K + 1500, max(4000, K + 700)])) if tile > 1 else int(rng.choice([K, K + 40]))
                text = naive.random_text(rng, first + L, p_amb=0.05 if dst == 4 and src != 2 else 0.0)
                words = source_words(text, src)
                view = source_words(text[first:], src)
                seq = cap.Seq(words.ctypes.data, L, first, 0, src, 0)
                n = L - K + 1
                res = cap.Result()
                tag = (src, dst, K, tile, first, L)
                efw, erv, _ = orc.fwrv(view, L, src, dst, K)
                ek, eh, _ = orc.canonical(view, L, src, dst, K, seed=5)
                fw, rv = np.zeros((n, N), np.uint64), np.zeros((n, N), np.uint64)
                assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(fw), vp(rv), cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
                assert np.array_equal(fw, efw) and np.array_equal(rv, erv), tag
                ck, hs = np.zeros((n, N), np.uint64), np.zeros(n, np.uint64)
                assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(ck), vp(hs), 5, cap.MEM_HOST, C.byref(res)) == 0
                assert np.array_equal(ck, ek) and np.array_equal(hs, eh), tag
                assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, None, vp(hs), 5, cap.MEM_HOST, C.byref(res)) == 0   # hashes only
                assert np.array_equal(hs, eh), tag
                t = np.zeros((n, 2 * N), np.uint64)
                assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, dst, vp(t), None, cap.OUT_TUPLES, C.byref(res)) == 0
                assert np.array_equal(t, np.concatenate([efw, erv], axis=1)), tag
                t = np.zeros((n, N + 1), np.uint64)
                assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(t), None, 5, cap.OUT_TUPLES, C.byref(res)) == 0
                assert np.array_equal(t, np.concatenate([ek, eh[:, None]], axis=1)), tag
                val = C.c_uint64()
                for canonical, e in ((0, efw), (1, ek)):
                    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, dst, canonical, C.byref(val), 0, C.byref(res)) == 0
                    assert val.value == int(np.bitwise_xor.reduce(e[:, 0])), tag
                out = np.zeros(40, np.uint64)
                assert ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, dst, 5, 40, vp(out), cap.MEM_HOST, C.byref(res)) == 0
                e = np.unique(eh)[:40]
                assert res.n_out == len(e) and np.array_equal(out[:len(e)], e), tag
                for J in (2, 33, K + 7):       # SpacedKmers: windows J symbols apart in the same staged stretch
                    es, _ = orc.spaced(view, L, src, dst, K, J)
                    sp = np.zeros((len(es), N), np.uint64)
                    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(sp), cap.MEM_HOST, C.byref(res)) == 0, ctx.last_error()
                    assert res.n_out == len(es) and np.array_equal(sp, es), tag + (J,)
                    assert ctx.lib.kmers_reduce_xor_iter(ctx.handle, C.byref(seq), K, dst, cap.ITER_SPACED, J, C.byref(val), 0, C.byref(res)) == 0
                    assert val.value == int(np.bitwise_xor.reduce(es[:, 0])), tag + (J,)
            if dst == 2 and src in (4, 8):
                ctx.set_param(cap.PARAM_TILE_KMERS, 64)
                L = max(3000, 3 * K + 400)
                t = list(naive.random_text(rng, L))
                t[2900], t[1234], t[1300] = "N", "R", "-"
                words = source_words("".join(t), src)
                seq = cap.Seq(words.ctypes.data, L, 0, 9, src, 0)    # index_origin 9: positions are reported behind it
                out = np.zeros((L - K + 1, N), np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, dst, vp(out), None, 0, cap.MEM_HOST, C.byref(res))
                assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, 1235 + 9, ord("R") if src == 8 else 0b0101)
                # SpacedKmers{K, K + 100}: windows [1, K], [K + 101, 2K + 100], ...; a symbol between two windows is never
                # inspected (SpacedKmers.jl:133-134), the first one inside a window is the error
                J = K + 100
                t = list(naive.random_text(rng, L))
                t[K + 49], t[J + 5], t[J + 2] = "N", "N", "S"       # 1-based K + 50: in the gap; J + 6 and J + 3: in the second window
                words = source_words("".join(t), src)
                seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                m = (L - K) // J + 1
                rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(out), cap.MEM_HOST, C.byref(res))
                _, eres = orc.spaced(words, L, src, dst, K, J)
                assert (rc, res.err_pos, res.err_enc) == (cap.E_ENCODE, J + 3, ord("S") if src == 8 else 0b0110) == (cap.E_ENCODE, eres.err_pos, eres.err_enc)
                t[J + 5], t[J + 2] = "A", "C"
                words = source_words("".join(t), src)
                seq = cap.Seq(words.ctypes.data, L, 0, 0, src, 0)
                es, eres = orc.spaced(words, L, src, dst, K, J)
                assert eres.status == 0
                assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, dst, vp(out), cap.MEM_HOST, C.byref(res)) == 0
                assert res.n_out == m and np.array_equal(out[:m], es)
    finally:
        ctx.close()
