"""kmers_batch: many records (ragged, some empty or shorter than K) in one launch == the per-record
iterators of the oracle, concatenated in record order; errors name the first failing record."""
import ctypes as C

import numpy as np
import pytest

import naive

pytestmark = pytest.mark.gpu


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


def build_pool(texts, src, rng, scatter):
    """Pool words + spans.  LongSequence pools place records at arbitrary symbol offsets (views of one
    long sequence, with filler between them when `scatter`); byte pools are the joined text."""
    import kmers_jl_amd as km
    spans, pieces, pos = [], [], 0
    for t in texts:
        gap = naive.random_text(rng, int(rng.integers(0, 40))) if scatter else ""
        pieces.append(gap + t)
        spans.append((pos + len(gap), len(t)))
        pos += len(gap) + len(t)
    whole = "".join(pieces)
    words = naive.ascii_words(whole) if src == 8 else naive.longseq_words(whole if whole else "A", src)
    arr = (km._capi.Span * max(len(spans), 1))(*[km._capi.Span(a, b) for a, b in spans])
    return words, arr, len(whole)


def expected(orc, texts, src, dst, K, mode, seed):
    outs_a, outs_b, offs = [], [], [0]
    for t in texts:
        if len(t) >= K:
            w = naive.ascii_words(t) if src == 8 else naive.longseq_words(t, src)
            if mode == 0:
                a, b, res = orc.fwrv(w, len(t), src, dst, K)
            else:
                a, b, res = orc.canonical(w, len(t), src, dst, K, seed=seed)
            assert res.status == 0
            outs_a.append(a)
            outs_b.append(b)
        offs.append(offs[-1] + max(0, len(t) - K + 1))
    N = (K * dst + 63) // 64
    a = np.concatenate(outs_a) if outs_a else np.zeros((0, N), np.uint64)
    b = np.concatenate(outs_b) if outs_b else np.zeros((0, N) if mode == 0 else (0,), np.uint64)
    return a, b, np.array(offs, np.uint64)


@pytest.mark.parametrize("src", [2, 4, 8])
def test_batch_matches_per_record_iteration(km, ctx, orc, src):
    cap = km._capi
    rng = np.random.default_rng(100 + src)
    for dst in (2, 4):
        for K in (1, 5, 31, 32, 33, 64, 65, 100, 128) if dst == 2 else (1, 7, 16, 17, 32, 33, 64):
            for n_rec, scatter in ((1, False), (7, True), (300, True), (4000, False)):
                lens = rng.choice([0, 1, K - 1, K, K + 1, 50, 151, 1000, 3000], n_rec)
                texts = [naive.random_text(rng, int(max(0, l))) for l in lens]
                if src == 8:
                    texts = ["".join(c.lower() if rng.random() < 0.3 else c for c in t) for t in texts]
                words, spans, n_pool = build_pool(texts, src, rng, scatter)
                seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
                for mode in (cap.BATCH_FW, cap.BATCH_CANONICAL):
                    ea, eb, eoff = expected(orc, texts, src, dst, K, mode, 9)
                    total = int(eoff[-1])
                    N = (K * dst + 63) // 64
                    res = cap.Result()
                    offs = np.zeros(n_rec + 1, np.uint64)
                    # size query
                    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, mode, K, dst, None, None, 9, vp(offs), 0, 0, C.byref(res))
                    assert rc == 0 and res.n_out == total and np.array_equal(offs, eoff), (src, dst, K, n_rec, mode)
                    out_a = np.zeros((max(total, 1), N), np.uint64)
                    out_b = np.zeros((max(total, 1), N) if mode == cap.BATCH_FW else max(total, 1), np.uint64)
                    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, mode, K, dst, vp(out_a), vp(out_b), 9, vp(offs),
                                             total, 0, C.byref(res))
                    assert rc == 0, ctx.last_error()
                    assert np.array_equal(out_a[:total], ea) and np.array_equal(out_b[:total], eb), (src, dst, K, n_rec, mode)
                    if total:  # too small a buffer is refused, with the required size
                        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, mode, K, dst, vp(out_a), None, 9, None,
                                                 total - 1, 0, C.byref(res))
                        assert rc == cap.E_CAPACITY and res.n_out == total


def test_batch_device_pointers_and_unaligned_outputs(km, ctx, orc):
    cap = km._capi
    rng = np.random.default_rng(5)
    K = 31
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(20, 400, 2000)]
    words, spans, n_pool = build_pool(texts, 4, rng, True)
    ea, eb, eoff = expected(orc, texts, 4, 2, K, cap.BATCH_CANONICAL, 0)
    total = int(eoff[-1])
    d_w = ctx.alloc(words.nbytes + 16)
    ctx.h2d(d_w, words)
    d_a, d_b = ctx.alloc(total * 8 + 32), ctx.alloc(total * 8 + 32)
    seq = cap.Seq(d_w, n_pool, 0, 0, 4, 0)
    res = cap.Result()
    for shift in (0, 8):
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, d_a + shift, d_b + shift, 0,
                                 None, total, cap.MEM_DEVICE, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        a, b = np.zeros(total, np.uint64), np.zeros(total, np.uint64)
        ctx.d2h(a, d_a + shift)
        ctx.d2h(b, d_b + shift)
        assert np.array_equal(a, ea[:, 0]) and np.array_equal(b, eb), shift
    for d in (d_w, d_a, d_b):
        ctx.free(d)


def test_batch_encode_error_names_record_and_position(km, ctx, orc):
    cap = km._capi
    rng = np.random.default_rng(6)
    K = 21
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(0, 300, 500)]
    # ambiguity in records too short to be iterated is never inspected (FwKmers.jl:63)
    texts[3] = "ACGTN"
    # two bad records: the earlier one (in batch order) wins, at its first offending symbol
    bad_late, bad_early = 400, 123
    texts[bad_late] = naive.random_text(rng, 100) + "W" + naive.random_text(rng, 50)
    texts[bad_early] = naive.random_text(rng, 77) + "R" + naive.random_text(rng, 10) + "N" + naive.random_text(rng, 30)
    for src in (4, 8):
        words, spans, n_pool = build_pool(texts, src, rng, True)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, None, None, 0, None, 0, 0,
                                 C.byref(res))
        total = int(res.n_out)
        out = np.zeros(total, np.uint64)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out), None, 0, None, total,
                                 0, C.byref(res))
        assert rc == cap.E_ENCODE and res.n_out == bad_early and res.err_pos == 78
        assert res.err_enc == (naive.DNA4["R"] if src == 4 else ord("R"))
        # the same pool with 4-bit kmers keeps the ambiguous symbols (Copyable / ascii_encode of a 4-bit alphabet)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_FW, K, 4, None, None, 0, None, 0, 0, C.byref(res))
        assert rc == 0
    # the next call starts clean
    texts[bad_late] = texts[bad_early] = "ACGT" * 30
    words, spans, n_pool = build_pool(texts, 4, rng, False)
    seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, 4, 0)
    ea, _, eoff = expected(orc, texts, 4, 2, K, cap.BATCH_CANONICAL, 0)
    out = np.zeros(int(eoff[-1]), np.uint64)
    assert ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out), None, 0, None, len(out),
                               0, C.byref(res)) == 0
    assert np.array_equal(out, ea[:, 0])


def test_collect_batch_mirror(km, orc):
    rng = np.random.default_rng(8)
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(0, 200, 300)]
    kmers, hashes, offs = km.collect_batch(km.CanonicalDNAMers[15], [km.LongDNA[4](t) for t in texts], hashes=True, seed=2)
    for i in (0, 17, 299):
        exp = naive.canonical(texts[i], 15, 2)
        got = kmers.words[offs[i]:offs[i + 1]]
        assert [tuple(int(x) for x in r) for r in got] == exp
        assert [int(h) for h in hashes[offs[i]:offs[i + 1]]] == [naive.fx_hash(w, 2) for w in exp]
    fw, rv, offs = km.collect_batch(km.FwRvIterator[km.DNAAlphabet[2], 9], texts)   # String records
    i = 5
    assert [(tuple(int(x) for x in a), tuple(int(x) for x in b)) for a, b in zip(fw.words[offs[i]:offs[i + 1]], rv.words[offs[i]:offs[i + 1]])] \
        == naive.fwrv(texts[i], 9, 2)
    with pytest.raises(km.EncodeError):
        km.collect_batch(km.FwDNAMers[4], ["ACGTACGT", "ACGNACGT"])
    k, _, offs = km.collect_batch(km.FwDNAMers[4], [])
    assert len(k) == 0 and list(offs) == [0]


@pytest.mark.parametrize("src", [2, 4, 8])
def test_batch_spaced_matches_per_record_iteration(km, ctx, orc, src):
    """kmers_batch_spaced == SpacedKmers{A,K,J}(record) record by record (SpacedKmers.jl:38-42,92-139): J < K (shift-ins),
    J == K (each_codon), J > K (gaps that are never inspected), one- and two-word kmers, 2- and 4-bit kmer alphabets."""
    cap = km._capi
    rng = np.random.default_rng(300 + src)
    for passes in (1, 8):
        ctx.set_param(cap.PARAM_BATCH_PASSES, passes)
        for dst, K, J in ((2, 3, 3), (2, 21, 3), (2, 5, 9), (2, 40, 7), (4, 7, 2), (4, 20, 20), (2, 1, 1)):
            for n_rec, scatter in ((1, False), (9, True), (1500, False)):
                lens = rng.choice([0, 1, K - 1, K, K + 1, K + J - 1, K + J, 60, 301, 2000], n_rec)
                texts = [naive.random_text(rng, int(max(0, l))) for l in lens]
                words, spans, n_pool = build_pool(texts, src, rng, scatter and src != 8)
                seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
                exp, eoff = [], [0]
                for t in texts:
                    if len(t) >= K:
                        w = naive.ascii_words(t) if src == 8 else naive.longseq_words(t, src)
                        e, r = orc.spaced(w, len(t), src, dst, K, J)
                        assert r.status == 0
                        exp.append(e)
                    eoff.append(eoff[-1] + (0 if len(t) < K else (len(t) - K) // J + 1))
                N = (K * dst + 63) // 64
                exp = np.concatenate(exp) if exp else np.zeros((0, N), np.uint64)
                total = eoff[-1]
                res = cap.Result()
                offs = np.zeros(n_rec + 1, np.uint64)
                rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, n_rec, K, J, dst, None, vp(offs), 0, 0, C.byref(res))
                assert rc == 0 and res.n_out == total and list(offs) == eoff, (src, dst, K, J, n_rec)
                out = np.zeros((max(total, 1), N), np.uint64)
                rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, n_rec, K, J, dst, vp(out), vp(offs), total, 0, C.byref(res))
                assert rc == 0 and res.n_out == total, ctx.last_error()
                assert np.array_equal(out[:total], exp), (src, dst, K, J, n_rec, passes)
    ctx.set_param(cap.PARAM_BATCH_PASSES, 0)
    if src == 2:
        return
    # strictness: a window over an ambiguous symbol fails with the record and the position; a gap symbol between
    # windows (J > K) is never inspected (test/runtests.jl:866-869)
    texts = ["ACGTACGTACGT", "TAGAWWWW", "ACGT"]
    words, spans, n_pool = build_pool(texts, src, rng, False)
    seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
    res = cap.Result()
    out = np.zeros((16, 1), np.uint64)
    rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, 3, 3, 4, 2, vp(out), None, 16, 0, C.byref(res))
    assert rc == cap.E_ENCODE and res.n_out == 1 and res.err_pos == 5 and res.err_enc == (ord("W") if src == 8 else 0b1001)
    texts = ["ACGNACGNACG", "TTTNTTT"]                       # K = 3, J = 4: every N falls between two windows
    words, spans, n_pool = build_pool(texts, src, rng, False)
    seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
    rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, 2, 3, 4, 2, vp(out), None, 16, 0, C.byref(res))
    assert rc == 0 and res.n_out == 5
    assert [int(x) for x in out[:5, 0]] == [naive.kmer_words(t, 2)[0] for t in ("ACG", "ACG", "ACG", "TTT", "TTT")]
    # skip mode marks instead of failing
    texts = ["ACGTNCGTACGT"]
    words, spans, n_pool = build_pool(texts, src, rng, False)
    seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
    rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, 1, 4, 2, 2, vp(out), None, 16, cap.BATCH_SKIP, C.byref(res))
    assert rc == 0 and res.n_out == 5
    want = [naive.kmer_words(t, 2)[0] if "N" not in t else 0xFFFFFFFFFFFFFFFF for t in ("ACGT", "GTNC", "NCGT", "GTAC", "ACGT")]
    assert [int(x) for x in out[:5, 0]] == want
    # the mirror: each_codon of every gene
    genes = ["ATGGCCTAA", "ATGTTTGGGCCCTGA", "AT"]
    kmers, none, offs = km.collect_batch(km.SpacedDNAMers[3, 3], [km.LongDNA[4](g) for g in genes] if src == 4 else genes)
    assert none is None and list(offs) == [0, 3, 8, 8]
    assert [str(k) for k in kmers] == ["ATG", "GCC", "TAA", "ATG", "TTT", "GGG", "CCC", "TGA"]


@pytest.mark.parametrize("src", [2, 4, 8])
def test_batch_pool_is_itself_a_view(km, ctx, orc, src):
    """pool.first_base != 0: the pool is a LongSubSeq of a longer buffer; spans are relative to the view."""
    cap = km._capi
    rng = np.random.default_rng(40 + src)
    K = 21
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(0, 120, 400)]
    for lead in (1, 5, 16, 33, 70):
        prefix = naive.random_text(rng, lead)
        whole = prefix + "".join(texts)
        words = naive.ascii_words(whole) if src == 8 else naive.longseq_words(whole, src)
        pos, spans = 0, []
        for t in texts:
            spans.append((pos, len(t)))
            pos += len(t)
        arr = (cap.Span * len(spans))(*[cap.Span(a, b) for a, b in spans])
        seq = cap.Seq(words.ctypes.data, pos, lead, 0, src, 0)
        ea, eb, eoff = expected(orc, texts, src, 2, K, cap.BATCH_CANONICAL, 4)
        total = int(eoff[-1])
        out_a, out_b = np.zeros(total, np.uint64), np.zeros(total, np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), arr, len(spans), cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 4, None,
                                 total, 0, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        assert np.array_equal(out_a, ea[:, 0]) and np.array_equal(out_b, eb), (src, lead)
        # a span that leaves the view is refused
        bad = (cap.Span * 1)(cap.Span(pos - 3, 10))
        assert ctx.lib.kmers_batch(ctx.handle, C.byref(seq), bad, 1, cap.BATCH_CANONICAL, K, 2, None, None, 0, None, 0, 0,
                                   C.byref(res)) == cap.E_BADARG


@pytest.mark.parametrize("src", [2, 4, 8])
def test_minhash_batch_matches_per_record_sketches(km, ctx, orc, src):
    """kmers_minhash_batch: record i's sketch == the s smallest distinct canonical hashes of record i."""
    cap = km._capi
    rng = np.random.default_rng(60 + src)
    for K, s in ((5, 10), (16, 100), (21, 1000), (31, 2048), (40, 64)):
        lens = rng.choice([0, K - 1, K, 50, 300, 5000, 40_000], 150)
        texts = [naive.random_text(rng, int(max(0, l))) for l in lens]
        texts[7] = "A" * 3000 + "ACGT" * 500          # low complexity: few distinct kmers
        words, spans, n_pool = build_pool(texts, src, rng, True)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        out = np.zeros((len(texts), s), np.uint64)
        counts = np.zeros(len(texts), np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, len(texts), K, 2, 3, s, vp(out), vp(counts), 0, C.byref(res))
        assert rc == 0 and res.n_out == len(texts), ctx.last_error()
        for i, t in enumerate(texts):
            if len(t) >= K:
                w = naive.ascii_words(t) if src == 8 else naive.longseq_words(t, src)
                _, eh, _ = orc.canonical(w, len(t), src, 2, K, seed=3)
                exp = np.unique(eh)[:s]
            else:
                exp = np.zeros(0, np.uint64)
            assert counts[i] == len(exp) and np.array_equal(out[i, :len(exp)], exp), (src, K, s, i, len(t))
    # errors name the record; sizes beyond the LDS sort are refused
    texts = ["ACGTACGTACGTACGTACGT", "ACGTACGTNACGTACGTACGTACGT"]
    if src != 2:
        words, spans, n_pool = build_pool(texts, src, rng, False)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, 2, 5, 2, 0, 10, vp(out), vp(counts), 0, C.byref(res))
        assert rc == cap.E_ENCODE and res.n_out == 1 and res.err_pos == 9
    assert ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, 2, 5, 2, 0, 5000, vp(out), vp(counts), 0, C.byref(res)) == cap.E_UNSUPPORTED


def test_sketch_batch_mirror(km, orc):
    rng = np.random.default_rng(61)
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(0, 3000, 40)]
    sk = km.sketch_batch(km.fx_hash, km.CanonicalDNAMers[16], [km.LongDNA[2](t) for t in texts], 50)
    for i, t in enumerate(texts):
        one = km.sketch(km.fx_hash, km.CanonicalDNAMers[16](km.LongDNA[2](t)), 50) if len(t) >= 16 else np.zeros(0, np.uint64)
        assert np.array_equal(sk[i], one), i


def test_batch_tiles_crowded_with_empty_records(km, ctx, orc):
    """Thousands of consecutive records that own nothing (shorter than K) inside one tile: the tile's slice of
    the layout does not fit LDS and the lanes search the global arrays instead."""
    cap = km._capi
    rng = np.random.default_rng(90)
    K = 25
    texts = []
    for block in range(6):
        texts += [naive.random_text(rng, int(l)) for l in rng.integers(K, 400, 5)]
        texts += [naive.random_text(rng, int(l)) for l in rng.integers(0, K, 3000)]   # own nothing
    texts += [naive.random_text(rng, 5000)]
    for src in (2, 4):
        words, spans, n_pool = build_pool(texts, src, rng, False)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        ea, eb, eoff = expected(orc, texts, src, 2, K, cap.BATCH_CANONICAL, 1)
        total = int(eoff[-1])
        out_a, out_b = np.zeros(total, np.uint64), np.zeros(total, np.uint64)
        offs = np.zeros(len(texts) + 1, np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 1, vp(offs),
                                 total, 0, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        assert np.array_equal(offs, eoff) and np.array_equal(out_a, ea[:, 0]) and np.array_equal(out_b, eb), src
        sk = np.zeros((len(texts), 20), np.uint64)
        cnt = np.zeros(len(texts), np.uint64)
        assert ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, len(texts), K, 2, 1, 20, vp(sk), vp(cnt), 0, C.byref(res)) == 0
        for i in (0, 4, 5, 3004, len(texts) - 1):
            n_i = int(eoff[i + 1] - eoff[i])
            e = np.unique(eb[int(eoff[i]):int(eoff[i + 1])])[:20]
            assert cnt[i] == len(e) and np.array_equal(sk[i, :len(e)], e), (src, i, n_i)


@pytest.mark.parametrize("passes", [1, 3, 8, 16])
def test_batch_tile_sizes_and_record_orders(km, ctx, orc, passes):
    """The tile (1024 elements x 1..8 passes, normally chosen from the batch size) forced to each extreme, over the
    layouts that take different paths of the element kernel: reads in pool order (windows cut from the staged
    stretch), the same records listed in shuffled order (windows outside the stretch: HBM), very short reads
    (more record slots than LDS holds: global search), one- and two-word kmers, strict and skip mode."""
    cap = km._capi
    ctx.set_param(cap.PARAM_BATCH_PASSES, passes)
    rng = np.random.default_rng(700 + passes)
    layouts = {
        "reads": [naive.random_text(rng, int(l)) for l in rng.integers(100, 260, 260)],
        "short": [naive.random_text(rng, int(l)) for l in rng.integers(28, 45, 2500)],
        "contigs": [naive.random_text(rng, int(l)) for l in (9000, 40, 17000, 5, 12000)],
    }
    for name, texts in layouts.items():
        for src, K, mode in ((2, 31, cap.BATCH_CANONICAL), (4, 31, cap.BATCH_CANONICAL), (4, 40, cap.BATCH_FW), (8, 12, cap.BATCH_FW)):
            for shuffled in (False, True):
                words, spans, n_pool = build_pool(texts, src, rng, scatter=(src != 8))
                order = rng.permutation(len(texts)) if shuffled else np.arange(len(texts))
                listed = [texts[i] for i in order]
                sp = (cap.Span * len(texts))(*[spans[int(i)] for i in order])
                seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
                ea, eb, eoff = expected(orc, listed, src, 2, K, mode, 3)
                total = int(eoff[-1])
                N = (K * 2 + 63) // 64
                out_a = np.zeros((total, N), np.uint64)
                out_b = np.zeros((total, N) if mode == cap.BATCH_FW else total, np.uint64)
                offs = np.zeros(len(texts) + 1, np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), sp, len(texts), mode, K, 2, vp(out_a), vp(out_b), 3, vp(offs), total, 0,
                                         C.byref(res))
                assert rc == 0 and res.n_out == total, ctx.last_error()
                assert np.array_equal(offs, eoff), (name, src, K, shuffled)
                assert np.array_equal(out_a, ea) and np.array_equal(out_b, eb), (name, src, K, shuffled)
    # skip mode: the same elements with the windows over an N written as all-ones
    texts = [naive.random_text(rng, int(l), p_amb=0.004) for l in rng.integers(100, 400, 200)]
    for src in (4, 8):
        words, spans, n_pool = build_pool(texts, src, rng, scatter=False)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        total = sum(max(0, len(t) - 31 + 1) for t in texts)
        out_a, out_b = np.zeros(total, np.uint64), np.zeros(total, np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, 31, 2, vp(out_a), vp(out_b), 0, None, total,
                                 cap.BATCH_SKIP, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        g = 0
        for t in texts:
            for j in range(max(0, len(t) - 31 + 1)):
                w = t[j:j + 31]
                if any(c not in "ACGT" for c in w):
                    assert out_a[g] == out_b[g] == 0xFFFFFFFFFFFFFFFF, (src, g)
                elif j % 7 == 0:   # (every seventh: the oracle call is per window here)
                    ek, eh, _ = orc.canonical(naive.ascii_words(w), 31, 8, 2, 31)
                    assert out_a[g] == ek[0, 0] and out_b[g] == eh[0], (src, g)
                g += 1
    ctx.set_param(cap.PARAM_BATCH_PASSES, 0)


@pytest.mark.parametrize("src", [4, 8])
def test_batch_long_kmers_see_every_symbol(km, ctx, orc, src):
    """Windows of more than 64 symbols (three- and four-word kmers) span up to three words of the flag stream: an
    ambiguous symbol anywhere in the window -- the record's last symbol included -- fails the call (strict) or marks
    exactly the windows over it (skip)."""
    cap = km._capi
    rng = np.random.default_rng(40 + src)
    for K in (65, 100, 128):
        texts = []
        for i in range(60):
            t = list(naive.random_text(rng, int(rng.integers(K, K + 200))))
            if i >= 7:                                   # the first records are clean
                t[len(t) - 1 if i % 5 == 0 else int(rng.integers(0, len(t)))] = "N"
            texts.append("".join(t))
        words, spans, n_pool = build_pool(texts, src, rng, scatter=(src != 8))
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        total = sum(len(t) - K + 1 for t in texts)
        N = (2 * K + 63) // 64
        out_a, out_b = np.zeros((total, N), np.uint64), np.zeros(total, np.uint64)
        res = cap.Result()
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 0, None, total, 0,
                                 C.byref(res))
        assert rc == cap.E_ENCODE and res.n_out == 7 and res.err_pos == texts[7].index("N") + 1, (K, res.n_out, res.err_pos)
        assert res.err_enc == (ord("N") if src == 8 else 0b1111)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 0, None, total,
                                 cap.BATCH_SKIP, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        g = 0
        for i, t in enumerate(texts):
            n_i = len(t) - K + 1
            marked = np.array(["N" in t[j:j + K] for j in range(n_i)])
            got = out_a[g:g + n_i]
            assert np.array_equal((got == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1), marked), (K, i)
            assert np.array_equal(out_b[g:g + n_i] == np.uint64(0xFFFFFFFFFFFFFFFF), marked), (K, i)
            if not marked.all():
                j = int(np.flatnonzero(~marked)[0])
                ek, eh, _ = orc.canonical(naive.ascii_words(t[j:j + K]), K, 8, 2, K)
                assert np.array_equal(got[j], ek[0]) and out_b[g + j] == eh[0], (K, i, j)
            g += n_i
        # per-record sketches leave the marked windows out
        sk = np.zeros((len(texts), 16), np.uint64)
        cnt = np.zeros(len(texts), np.uint64)
        assert ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, len(texts), K, 2, 0, 16, vp(sk), vp(cnt), cap.BATCH_SKIP, C.byref(res)) == 0
        g = 0
        for i, t in enumerate(texts):
            n_i = len(t) - K + 1
            h = out_b[g:g + n_i]
            e = np.unique(h[h != np.uint64(0xFFFFFFFFFFFFFFFF)])[:16]
            assert cnt[i] == len(e) and np.array_equal(sk[i, :len(e)], e), (K, i)
            g += n_i


@pytest.mark.parametrize("src", [4, 8])
def test_batch_skip_mode_masks_ambiguous_windows(km, ctx, orc, src):
    """KMERS_BATCH_SKIP: reads with N do not fail; the elements whose window holds an ambiguous symbol are
    all-ones, the others equal the strict result -- i.e. the kept ones are UnambiguousKmers of each record
    (checked through the oracle's UnambiguousKmers: kept indices == its start positions)."""
    cap = km._capi
    rng = np.random.default_rng(70 + src)
    for K in (5, 21, 31, 33):
        texts = [naive.random_text(rng, int(l), p_amb=0.02) for l in rng.choice([0, K - 1, K, 60, 150, 151, 2000], 400)]
        words, spans, n_pool = build_pool(texts, src, rng, True)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        N = (2 * K + 63) // 64
        offs = np.zeros(len(texts) + 1, np.uint64)
        res = cap.Result()
        assert ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, None, None, 3, vp(offs), 0,
                                   cap.BATCH_SKIP, C.byref(res)) == 0
        total = int(res.n_out)
        out_a, out_b = np.zeros((total, N), np.uint64), np.zeros(total, np.uint64)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, vp(out_a), vp(out_b), 3, None, total,
                                 cap.BATCH_SKIP, C.byref(res))
        assert rc == 0, ctx.last_error()
        sk = np.zeros((len(texts), 50), np.uint64)
        cnt = np.zeros(len(texts), np.uint64)
        assert ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, len(texts), K, 2, 3, 50, vp(sk), vp(cnt), cap.BATCH_SKIP,
                                           C.byref(res)) == 0
        for i, t in enumerate(texts):
            lo, hi = int(offs[i]), int(offs[i + 1])
            assert hi - lo == max(0, len(t) - K + 1)
            if hi == lo:
                assert cnt[i] == 0
                continue
            w = naive.ascii_words(t) if src == 8 else naive.longseq_words(t, src)
            uk, us, _ = orc.unambiguous(w, len(t), src, K)            # kept windows: (forward kmer, 1-based start)
            kept = np.zeros(hi - lo, bool)
            kept[us - 1] = True
            got = out_a[lo:hi]
            assert np.array_equal(~(got == np.uint64(0xFFFFFFFFFFFFFFFF)).all(axis=1), kept), (src, K, i)
            canon = np.array([orc.canonical_kmer(tuple(int(x) for x in r), K, 2) for r in uk], dtype=np.uint64).reshape(-1, N)
            assert np.array_equal(got[kept], canon), (src, K, i)
            hashes = np.array([orc.fx_hash(tuple(int(x) for x in r), 3) for r in canon], dtype=np.uint64)
            assert np.array_equal(out_b[lo:hi][kept], hashes) and (out_b[lo:hi][~kept] == np.uint64(0xFFFFFFFFFFFFFFFF)).all()
            e = np.unique(hashes)[:50]
            assert cnt[i] == len(e) and np.array_equal(sk[i, :len(e)], e), (src, K, i)


def test_collect_batch_skip_ambiguous(km):
    reads = ["ACGTACGTAC", "ACGTNACGTACG", "NNNN", "ACGTACGTACGTTT"]
    with pytest.raises(km.EncodeError):
        km.collect_batch(km.CanonicalDNAMers[4], reads)
    kmers, _, offs = km.collect_batch(km.CanonicalDNAMers[4], reads, skip_ambiguous=True)
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    rec1 = kmers.words[offs[1]:offs[2], 0]
    assert list(rec1 == ones) == [False, True, True, True, True] + [False] * 4   # windows 2..5 cover the N
    assert (kmers.words[offs[2]:offs[3], 0] == ones).all() and offs[3] - offs[2] == 1
    assert not (kmers.words[offs[0]:offs[1], 0] == ones).any() and not (kmers.words[offs[3]:offs[4], 0] == ones).any()


@pytest.mark.parametrize("src", [2, 4, 8])
def test_batch_dense_tiles_equal_the_general_path_and_the_oracle(km, ctx, orc, src):
    """The dense tile path of ragged_kernel (csrc/ragged_kernels.hpp: record lookup by bitmap, one window cut per run, sub-runs at
    record boundaries from a list) against the oracle AND against the general path of the same library (KMERS_PARAM_BATCH_DENSE =
    -1): reads long enough for every record to own RG_RUN elements, in pool order with and without filler between them, tile
    lengths of 1, 3 and 8 passes, batches that end inside a run, and batches in which a short or empty record here and there sends
    single tiles back to the general path."""
    cap = km._capi
    rng = np.random.default_rng(500 + src)
    cases = []
    for dst, K in ((2, 31), (2, 32), (2, 5), (4, 16), (4, 9)):
        for n_rec, lo, hi, scatter, spoil in ((3000, K + 3, K + 4, False, 0), (2500, K + 3, 160, True, 0), (1800, 100, 400, False, 0),
                                              (2500, K + 3, 160, False, 40), (40, 3000, 9000, True, 0), (1, 70000, 70001, False, 0)):
            cases.append((dst, K, n_rec, lo, hi, scatter, spoil))
    for dst, K, n_rec, lo, hi, scatter, spoil in cases:
        lens = rng.integers(lo, hi, n_rec)
        for i in rng.integers(0, n_rec, spoil):          # records that own fewer than four elements, or none
            lens[i] = int(rng.choice([0, K - 1, K, K + 2]))
        texts = [naive.random_text(rng, int(l)) for l in lens]
        if src == 8:
            texts = ["".join(c.lower() if rng.random() < 0.3 else c for c in t) for t in texts]
        words, spans, n_pool = build_pool(texts, src, rng, scatter)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        for mode in (cap.BATCH_FW, cap.BATCH_CANONICAL):
            ea, eb, eoff = expected(orc, texts, src, dst, K, mode, 9)
            total = int(eoff[-1])
            got = {}
            for dense in (0, -1):
                ctx.set_param(cap.PARAM_BATCH_DENSE, dense)
                ctx.set_param(cap.PARAM_BATCH_PASSES, int(rng.choice([0, 1, 3, 8, 16])))
                a = np.full((max(total, 1), 1), 0xAAAAAAAAAAAAAAAA, np.uint64)
                b = np.full((max(total, 1), 1) if mode == cap.BATCH_FW else max(total, 1), 0xBBBBBBBBBBBBBBBB, np.uint64)
                off = np.zeros(n_rec + 1, np.uint64)
                res = cap.Result()
                rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, mode, K, dst, vp(a), vp(b), 9, vp(off), total, 0, C.byref(res))
                tag = (src, dst, K, n_rec, lo, hi, scatter, spoil, mode, dense)
                assert rc == 0 and res.n_out == total and np.array_equal(off, eoff), tag + (ctx.last_error(),)
                assert np.array_equal(a[:total], ea) and np.array_equal(b[:total], eb), tag
                got[dense] = (a, b)
            assert np.array_equal(got[0][0], got[-1][0]) and np.array_equal(got[0][1], got[-1][1])
    ctx.set_param(cap.PARAM_BATCH_DENSE, 0)
    ctx.set_param(cap.PARAM_BATCH_PASSES, 0)


@pytest.mark.parametrize("src", [4, 8])
def test_batch_of_reads_with_ns_on_the_dense_path(km, ctx, orc, src):
    """Reads as people have them (round 6): text or a 4-bit pool, lengths 50-250, an N in one read out of a hundred or out of ten
    (docs/src/faq.md:28-33; src/iterators/UnambiguousKmers.jl:109-148).  A symbol the kmer alphabet cannot encode costs the ELEMENTS
    whose windows hold it, not its tile: the dense tile path (csrc/ragged_kernels.hpp, dense_runs<FLAGGED>) writes the all-ones
    sentinel itself (KMERS_BATCH_SKIP) or reports the first failing element (strict) -- from the recoded stream and, in the
    optimistic launch, from the pool itself.  Against the oracle and against the general path (KMERS_PARAM_BATCH_DENSE = -1), host
    and device outputs, device outputs with the count deferred to the end of the call."""
    cap = km._capi
    rng = np.random.default_rng(900 + src)
    ones = np.uint64(0xFFFFFFFFFFFFFFFF)
    for K, dst, p_read in ((31, 2, 0.01), (31, 2, 0.10), (21, 2, 0.10), (9, 4, 0.05)):
        n_rec = 6000
        texts = []
        for l in rng.integers(50, 251, n_rec):
            t = naive.random_text(rng, int(l))
            if rng.random() < p_read:                                   # one to three Ns somewhere in the read
                t = list(t)
                for pos in rng.integers(0, len(t), int(rng.integers(1, 4))):
                    t[pos] = "N"
                t = "".join(t)
            if src == 8:
                t = "".join(c.lower() if rng.random() < 0.2 else c for c in t)
            texts.append(t)
        words, spans, n_pool = build_pool(texts, src, rng, False)
        seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
        N = 1
        # what every element should be: the strict result of the clean windows, all-ones where a window holds an N
        exp_a, exp_b, offs, first_bad = [], [], [0], None
        for i, t in enumerate(texts):
            n = max(0, len(t) - K + 1)
            up = t.upper()
            clean = up.replace("N", "A")
            w = naive.ascii_words(clean) if src == 8 else naive.longseq_words(clean, src)
            if dst == 2:
                a, b, r = orc.canonical(w, len(t), src, dst, K, seed=5)
            else:
                a, b, r = orc.canonical(w, len(t), src, dst, K, seed=5)
            assert r.status == 0
            holds = np.array([("N" in up[j:j + K]) for j in range(n)], bool)
            if dst == 4:                                                # a 4-bit kmer alphabet encodes N: nothing is masked, nothing fails
                w2 = naive.ascii_words(t) if src == 8 else naive.longseq_words(up, src)
                a, b, r = orc.canonical(w2, len(t), src, dst, K, seed=5)
                holds[:] = False
            a, b = a.copy(), b.copy()
            a[holds] = ones
            b[holds] = ones
            if holds.any() and first_bad is None:
                first_bad = (i, up.index("N") + 1)                     # the reference throws at the record's first N (1-based)
            exp_a.append(a)
            exp_b.append(b)
            offs.append(offs[-1] + n)
        exp_a, exp_b, total = np.concatenate(exp_a), np.concatenate(exp_b), offs[-1]
        res = cap.Result()
        got = {}
        for dense in (0, -1):
            ctx.set_param(cap.PARAM_BATCH_DENSE, dense)
            a, b = np.zeros((total, N), np.uint64), np.zeros(total, np.uint64)
            rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, cap.BATCH_CANONICAL, K, dst, vp(a), vp(b), 5, None, total,
                                     cap.BATCH_SKIP, C.byref(res))
            assert rc == 0 and res.n_out == total, (src, K, dense, ctx.last_error())
            assert np.array_equal(a, exp_a) and np.array_equal(b, exp_b), (src, K, dst, p_read, dense)
            got[dense] = a
            # strict: the first failing record and the position of its first N
            rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, n_rec, cap.BATCH_CANONICAL, K, dst, vp(a), vp(b), 5, None, total, 0, C.byref(res))
            if first_bad is None:
                assert rc == 0
            else:
                assert rc == cap.E_ENCODE and (int(res.n_out), int(res.err_pos)) == first_bad, (src, K, dense, int(res.n_out), int(res.err_pos), first_bad)
        # device outputs: the count is read on the device, the call waits once (the optimistic launch) -- same elements
        ctx.set_param(cap.PARAM_BATCH_DENSE, 0)
        d_w, d_a, d_b = ctx.alloc(words.nbytes + 16), ctx.alloc(total * 8 + 16), ctx.alloc(total * 8 + 16)
        ctx.h2d(d_w, words)
        dseq = cap.Seq(d_w, n_pool, 0, 0, src, 0)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(dseq), spans, n_rec, cap.BATCH_CANONICAL, K, dst, d_a, d_b, 5, None, total + 1000,
                                 cap.BATCH_SKIP | cap.MEM_DEVICE, C.byref(res))
        assert rc == 0 and res.n_out == total, ctx.last_error()
        a, b = np.zeros((total, N), np.uint64), np.zeros(total, np.uint64)
        ctx.d2h(a, d_a)
        ctx.d2h(b, d_b)
        assert np.array_equal(a, exp_a) and np.array_equal(b, exp_b), (src, K, "device outputs")
        for q in (d_w, d_a, d_b):
            ctx.free(q)
    ctx.set_param(cap.PARAM_BATCH_DENSE, 0)


def test_batch_with_device_outputs_waits_once_and_still_reports_everything(km, ctx, orc):
    """Device outputs and a sane capacity: the call sizes its layout for the CAPACITY and reads the element count on the device (no
    host round trip between the scan and the element kernel, csrc/batch_api.hip).  Same elements, the true count in res.n_out,
    nothing written beyond it; a capacity that is too small, a span outside the pool, an ambiguous symbol and a batch without any
    element are still reported as before."""
    cap = km._capi
    rng = np.random.default_rng(77)
    K = 31
    texts = [naive.random_text(rng, int(l)) for l in rng.integers(40, 300, 6000)]
    texts[17] = "ACGT"                                         # (a record shorter than K: one tile leaves the dense path)
    words, spans, n_pool = build_pool(texts, 4, rng, False)
    ea, eb, eoff = expected(orc, texts, 4, 2, K, cap.BATCH_CANONICAL, 5)
    total = int(eoff[-1])
    d_w = ctx.alloc(words.nbytes + 16)
    ctx.h2d(d_w, words)
    room = total + 70_000
    d_a, d_b = ctx.alloc(room * 8), ctx.alloc(room * 8)
    seq = cap.Seq(d_w, n_pool, 0, 0, 4, 0)
    res = cap.Result()
    sentinel = np.full(room, 0x5A5A5A5A5A5A5A5A, np.uint64)
    for capacity in (total, total + 1, room):                  # exact, and two upper bounds (a grid with tiles past the count)
        ctx.h2d(d_a, sentinel)
        ctx.h2d(d_b, sentinel)
        off = np.zeros(len(texts) + 1, np.uint64)
        rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, d_a, d_b, 5, vp(off), capacity,
                                 cap.MEM_DEVICE, C.byref(res))
        assert rc == 0 and res.n_out == total and np.array_equal(off, eoff), (capacity, ctx.last_error())
        a, b = np.zeros(room, np.uint64), np.zeros(room, np.uint64)
        ctx.d2h(a, d_a)
        ctx.d2h(b, d_b)
        assert np.array_equal(a[:total], ea[:, 0]) and np.array_equal(b[:total], eb), capacity
        assert np.all(a[total:] == sentinel[total:]) and np.all(b[total:] == sentinel[total:]), capacity
    # too small: the count needed comes back, nothing is written
    ctx.h2d(d_a, sentinel)
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(texts), cap.BATCH_CANONICAL, K, 2, d_a, d_b, 5, None, total - 1, cap.MEM_DEVICE,
                             C.byref(res))
    assert rc == cap.E_CAPACITY and res.n_out == total
    a = np.zeros(room, np.uint64)
    ctx.d2h(a, d_a)
    assert np.all(a == sentinel)
    # a span that reaches outside the pool
    bad = (cap.Span * len(texts))(*[cap.Span(s.first_base, s.n_bases) for s in spans])
    bad[100] = cap.Span(n_pool - 5, 50)
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), bad, len(texts), cap.BATCH_CANONICAL, K, 2, d_a, d_b, 5, None, room, cap.MEM_DEVICE, C.byref(res))
    assert rc == cap.E_BADARG and "outside the pool" in ctx.last_error()
    # an ambiguous symbol: the first failing record and the position inside it
    amb = list(texts)
    amb[4000] = amb[4000][:33] + "N" + amb[4000][34:]
    w2, s2, n2 = build_pool(amb, 4, rng, False)
    ctx.h2d(d_w, w2)
    seq2 = cap.Seq(d_w, n2, 0, 0, 4, 0)
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq2), s2, len(amb), cap.BATCH_CANONICAL, K, 2, d_a, d_b, 5, None, room, cap.MEM_DEVICE, C.byref(res))
    assert rc == cap.E_ENCODE and res.n_out == 4000 and res.err_pos == 34, (rc, res.n_out, res.err_pos)
    # no record long enough: nothing to do, status 0
    short = ["ACGTACGT"] * 50
    w3, s3, n3 = build_pool(short, 4, rng, False)
    ctx.h2d(d_w, w3)
    seq3 = cap.Seq(d_w, n3, 0, 0, 4, 0)
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq3), s3, len(short), cap.BATCH_CANONICAL, K, 2, d_a, d_b, 5, None, 1000, cap.MEM_DEVICE, C.byref(res))
    assert rc == 0 and res.n_out == 0
    for d in (d_w, d_a, d_b):
        ctx.free(d)


def test_batch_layout_of_millions_of_records_in_one_kernel(km, ctx):
    """The layout pass (csrc/scan_kernels.hpp: one kernel, segments of 4096 records chained by a decoupled look-back, descriptors
    tagged with the call's epoch instead of being cleared): offsets of up to three million ragged records -- many of them shorter
    than K -- equal the running sum of FwKmers.jl:40-43 / SpacedKmers.jl:38-42, call after call with batches that grow and shrink
    (stale descriptors of earlier calls lie where the next call looks back), at stride 1 and 3."""
    cap = km._capi
    rng = np.random.default_rng(2025)
    n_pool = 1 << 20
    words = np.zeros(n_pool * 2 // 64 + 2, np.uint64)
    seq = cap.Seq(words.ctypes.data, n_pool, 0, 0, 2, 0)
    res = cap.Result()
    K = 31
    for n, J in ((700_000, 1), (3_000_000, 1), (4097, 1), (3_000_000, 3), (1_000_001, 1), (8192, 3), (1, 1)):
        first = rng.integers(0, n_pool - 400, n).astype(np.uint64)
        length = rng.integers(0, 400, n).astype(np.uint64)
        length[rng.integers(0, n, n // 50)] = 0
        spans = np.stack([first, length], axis=1).copy()
        want = np.zeros(n + 1, np.uint64)
        cnt = np.where(length >= K, (length - np.uint64(K)) // np.uint64(J) + np.uint64(1), np.uint64(0))
        np.cumsum(cnt, out=want[1:])
        off = np.zeros(n + 1, np.uint64)
        sp = spans.ctypes.data_as(C.POINTER(cap.Span))
        if J == 1:
            rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), sp, n, cap.BATCH_FW, K, 2, None, None, 0, vp(off), 0, 0, C.byref(res))
        else:
            rc = ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), sp, n, K, J, 2, None, vp(off), 0, 0, C.byref(res))
        assert rc == 0 and res.n_out == int(want[-1]), (n, J, ctx.last_error())
        assert np.array_equal(off, want), (n, J, int(np.flatnonzero(off != want)[0]))
    # a span outside the pool, far into a large batch
    spans[n // 2, 0] = n_pool
    spans[n // 2, 1] = 1
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.ctypes.data_as(C.POINTER(cap.Span)), n, cap.BATCH_FW, K, 2, None, None, 0, None, 0, 0,
                             C.byref(res))
    assert rc == cap.E_BADARG and "outside the pool" in ctx.last_error()


def test_batch_layout_gives_up_instead_of_hanging(km):
    """The give-up path of the layout pass's look-back (scan_kernels.hpp: LAYOUT_SPIN_LIMIT, the abort word) cannot be provoked on a
    healthy device, so the TEST BUILD of the library provokes it (libkmers_hip_testabort.so, -DKMERS_TEST_ABORT: segment 1 never
    publishes anything).  A size query and a call with device outputs (which learns the verdict only at its end: the element kernel
    must have written nothing) both come back with KMERS_E_HIP, and the same context serves a batch of one segment right behind
    them.  A fresh process: one library per process."""
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
n, rl, K = 20_000, 100, 31                   # five segments of 4096 records: segments 2.. wait for segment 1
n_pool = n * rl
nw = n_pool // 16 + 1
d_w = ctx.alloc(nw * 8 + 8)
ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 3, 0, nw, 4, 0, d_w), "synth")
spans = np.stack([np.arange(n, dtype=np.uint64) * np.uint64(rl), np.full(n, rl, np.uint64)], axis=1).copy()
sp = spans.ctypes.data_as(C.POINTER(cap.Span))
total = n * (rl - K + 1)
d_a, d_b = ctx.alloc(total * 8), ctx.alloc(total * 8)
sentinel = np.full(total, 0x5A5A5A5A5A5A5A5A, np.uint64)
ctx.h2d(d_a, sentinel)
seq = cap.Seq(d_w, n_pool, 0, 0, 4, 0)
res = cap.Result()
rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), sp, n, cap.BATCH_CANONICAL, K, 2, None, None, 0, None, 0, cap.MEM_DEVICE, C.byref(res))
assert rc == cap.E_HIP and "gave up" in ctx.last_error(), (rc, ctx.last_error())
rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), sp, n, cap.BATCH_CANONICAL, K, 2, d_a, d_b, 0, None, total, cap.MEM_DEVICE, C.byref(res))
assert rc == cap.E_HIP and "gave up" in ctx.last_error(), (rc, ctx.last_error())
a = np.zeros(total, np.uint64)
ctx.d2h(a, d_a)
assert np.all(a == sentinel)
rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), sp, 4096, cap.BATCH_CANONICAL, K, 2, d_a, d_b, 0, None, total, cap.MEM_DEVICE, C.byref(res))
assert rc == 0 and res.n_out == 4096 * (rl - K + 1), (rc, ctx.last_error())
print("ok", res.n_out)
'''
    env = dict(os.environ, KMERS_HIP_LIB=lib, PYTHONPATH=os.pathsep.join(sys.path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.stdout, r.stderr[-2000:])


@pytest.mark.parametrize("src", [4, 8])
def test_minhash_batch_of_a_large_host_pool_comes_up_in_pieces(km, ctx, orc, src):
    """A pool of 128 MiB and more in HOST memory whose records lie in pool order is copied, recoded and sketched piece by piece
    (csrc/batch_api.hip, minhash_batch_fused: the copy of piece c + 1 beside the kernels of piece c).  Same sketches as the one-piece
    path (the same pool in device memory) for every record -- ragged, overlapping, with gaps, some shorter than K -- and as the oracle
    for a sample of them; an ambiguous symbol far into the pool is reported with its record and position."""
    cap = km._capi
    rng = np.random.default_rng(300 + src)
    K, s = 21, 200
    n_pool = (300 if src == 4 else 150) * (1 << 20)          # 150 MiB of source bytes: four pieces and a bit
    nw = (n_pool * src + 63) // 64
    d_w = ctx.alloc(nw * 8 + 16)
    if src == 4:
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, 11, 0, nw, 4, 0, d_w), "synth")
        words = np.zeros(nw + 2, np.uint64)
        ctx.d2h(words[:nw], d_w)
    else:
        words = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, nw * 8 + 16, dtype=np.uint8)].copy().view(np.uint64)
        ctx.h2d(d_w, words[:nw])
    n_rec = 3000
    first = np.sort(rng.integers(0, n_pool - 200_000, n_rec)).astype(np.uint64)
    length = rng.integers(0, 200_000, n_rec).astype(np.uint64)
    length[rng.integers(0, n_rec, 40)] = rng.integers(0, K, 40).astype(np.uint64)
    first[0] = 0
    spans = np.stack([first, length], axis=1).copy()
    sp = spans.ctypes.data_as(C.POINTER(cap.Span))
    res = cap.Result()
    out_h, cnt_h = np.zeros((n_rec, s), np.uint64), np.zeros(n_rec, np.uint64)
    seq_h = cap.Seq(words.ctypes.data, n_pool, 0, 0, src, 0)
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_h), sp, n_rec, K, 2, 9, s, vp(out_h), vp(cnt_h), 0, C.byref(res))
    assert rc == 0 and res.n_out == n_rec, ctx.last_error()
    assert ctx.last_batch_pieces() >= 4
    d_o, d_c = ctx.alloc(n_rec * s * 8), ctx.alloc(n_rec * 8)
    seq_d = cap.Seq(d_w, n_pool, 0, 0, src, 0)
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_d), sp, n_rec, K, 2, 9, s, d_o, d_c, cap.MEM_DEVICE, C.byref(res))
    assert rc == 0 and ctx.last_batch_pieces() == 1, ctx.last_error()
    out_d, cnt_d = np.zeros((n_rec, s), np.uint64), np.zeros(n_rec, np.uint64)
    ctx.d2h(out_d, d_o)
    ctx.d2h(cnt_d, d_c)
    assert np.array_equal(cnt_h, cnt_d)
    for i in range(n_rec):
        assert np.array_equal(out_h[i, :int(cnt_h[i])], out_d[i, :int(cnt_d[i])]), i
    for i in (0, 1, 17, n_rec // 2, n_rec - 2, n_rec - 1):                # the oracle, record by record
        f, l = int(first[i]), int(length[i])
        if src == 8:
            w = words.view(np.uint8)[f:f + l + 8].copy()
            w = np.concatenate([w, np.zeros((-len(w)) % 8, np.uint8)]).view(np.uint64)
            _, eh, _ = orc.canonical(w, l, 8, 2, K, seed=9)
        else:
            sh = f % 16                                                   # (records begin anywhere inside a word)
            w = words[f // 16:(f + l) // 16 + 2]
            if sh:
                w = (w[:-1] >> np.uint64(4 * sh)) | (w[1:] << np.uint64(64 - 4 * sh))
            _, eh, _ = orc.canonical(np.ascontiguousarray(w), l, 4, 2, K, seed=9)
        exp = np.unique(eh)[:s] if l >= K else np.zeros(0, np.uint64)
        assert cnt_h[i] == len(exp) and np.array_equal(out_h[i, :len(exp)], exp), i
    # an ambiguous symbol in the last piece: the record and the position inside it
    i_bad = n_rec - 5
    while int(length[i_bad]) < 1000:
        i_bad -= 1
    p_bad = int(first[i_bad]) + 777
    if src == 8:
        words.view(np.uint8)[p_bad] = ord("N")
    else:
        words[p_bad // 16] |= np.uint64(0xF) << np.uint64(4 * (p_bad % 16))
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_h), sp, n_rec, K, 2, 9, s, vp(out_h), vp(cnt_h), 0, C.byref(res))
    first_hit = min(i for i in range(n_rec) if int(first[i]) <= p_bad < int(first[i]) + int(length[i]) and int(length[i]) >= K)
    assert rc == cap.E_ENCODE and res.n_out == first_hit and res.err_pos == p_bad - int(first[first_hit]) + 1, (rc, res.n_out, res.err_pos, first_hit)
    # ... and with KMERS_BATCH_SKIP the windows over it are left out: pieces and one piece agree again (the flags of a piece are
    # written by that piece's recode pass)
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_h), sp, n_rec, K, 2, 9, s, vp(out_h), vp(cnt_h), cap.BATCH_SKIP, C.byref(res))
    assert rc == 0, ctx.last_error()
    ctx.h2d(d_w, words[:nw])
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq_d), sp, n_rec, K, 2, 9, s, d_o, d_c, cap.MEM_DEVICE | cap.BATCH_SKIP, C.byref(res))
    assert rc == 0, ctx.last_error()
    ctx.d2h(out_d, d_o)
    ctx.d2h(cnt_d, d_c)
    assert np.array_equal(cnt_h, cnt_d)
    for i in range(n_rec):
        assert np.array_equal(out_h[i, :int(cnt_h[i])], out_d[i, :int(cnt_d[i])]), i
    for d in (d_w, d_o, d_c):
        ctx.free(d)
