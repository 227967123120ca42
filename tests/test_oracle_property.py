"""Differential tests: rolling oracle (oracle/kmers_oracle.c) vs the independent naive
slicer (tests/naive.py), mirroring the reference's own property tests
(test/runtests.jl:438-485, :674-690, :739-761, :774-787, :805-847, :850-867)."""
import numpy as np
import pytest

import naive

KS = [1, 2, 3, 7, 15, 16, 21, 31, 32, 33, 36, 47, 63, 64, 65, 96, 128]


def rows(a):
    return [tuple(int(x) for x in r) for r in a]


@pytest.mark.parametrize("src", [2, 4])
@pytest.mark.parametrize("dst", [2, 4])
def test_fw_fwrv_canonical_random(orc, src, dst):
    rng = np.random.default_rng(100 + src * 10 + dst)
    for K in KS:
        if orc.nwords(K, dst) > 8:
            continue
        for L in (0, K - 1, K, K + 1, K + 40, 200):
            if L < 0:
                continue
            text = naive.random_text(rng, L)
            seq = naive.longseq_words(text, src)
            km, res = orc.fw_kmers(seq, L, src, dst, K)
            assert res.status == 0 and rows(km) == naive.fw_kmers(text, K, dst)
            fw, rv, res = orc.fwrv(seq, L, src, dst, K)
            assert res.status == 0
            assert list(zip(rows(fw), rows(rv))) == naive.fwrv(text, K, dst)
            ck, hs, res = orc.canonical(seq, L, src, dst, K, seed=K)
            exp = naive.canonical(text, K, dst)
            assert rows(ck) == exp
            assert [int(h) for h in hs] == [naive.fx_hash(w, K) for w in exp]
            x, _ = orc.reduce_xor_canonical(seq, L, src, dst, K)
            acc = 0
            for w in exp:
                acc ^= w[0]
            assert x == acc


@pytest.mark.parametrize("dst", [2, 4])
def test_four_bit_source_with_ambiguity(orc, dst):
    """4-bit source holding IUPAC codes: 2-bit kmers throw at the first inspected ambiguous
    symbol (FwKmers.jl:112, CanonicalKmers.jl:139); 4-bit kmers keep them (Copyable)."""
    rng = np.random.default_rng(7 + dst)
    for K in (1, 3, 21, 31, 33):
        for L in (K, K + 5, 150):
            text = naive.random_text(rng, L, p_amb=0.03)
            seq = naive.longseq_words(text, 4)
            km, res = orc.fw_kmers(seq, L, 4, dst, K)
            fw, rv, res2 = orc.fwrv(seq, L, 4, dst, K)
            if dst == 4:
                assert res.status == 0 and rows(km) == naive.fw_kmers(text, K, 4)
                assert list(zip(rows(fw), rows(rv))) == naive.fwrv(text, K, 4)
                continue
            bad = naive.first_ambiguous(text, range(1, L + 1))
            if bad is None:
                assert res.status == 0 and rows(km) == naive.fw_kmers(text, K, 2)
            else:
                assert res.status == 1 and res.err_pos == bad
                assert res.err_enc == naive.DNA4[text[bad - 1]]
                assert res2.status == 1 and res2.err_pos == bad
                # elements yielded before the throw: windows that end before the bad symbol
                assert res.n_out == max(0, bad - K)
                assert rows(km) == naive.fw_kmers(text[:res.n_out + K - 1], K, 2)


def test_unambiguous_random(orc):
    rng = np.random.default_rng(11)
    for K in (1, 2, 3, 5, 21, 31, 32, 33, 63):
        for L in (0, K - 1, K, K + 3, 100, 400):
            if L < 0:
                continue
            for p in (0.0, 0.04, 0.3):
                text = naive.random_text(rng, L, p_amb=p)
                km, st, res = orc.unambiguous(naive.longseq_words(text, 4), L, 4, K)
                assert res.status == 0
                assert list(zip(rows(km), [int(s) for s in st])) == naive.unambiguous(text, K)
            text = naive.random_text(rng, L)
            km, st, res = orc.unambiguous(naive.longseq_words(text, 2), L, 2, K)
            assert list(zip(rows(km), [int(s) for s in st])) == naive.unambiguous(text, K)


@pytest.mark.parametrize("src", [2, 4])
@pytest.mark.parametrize("dst", [2, 4])
def test_spaced_random(orc, src, dst):
    rng = np.random.default_rng(13 + src + 3 * dst)
    for K, J in [(3, 2), (2, 4), (3, 3), (4, 3), (21, 3), (31, 1), (31, 7), (33, 5), (5, 40), (31, 31), (31, 32)]:
        for L in (0, K - 1, K, K + 1, K + J, 3 * K + 2 * J + 1, 300):
            if L < 0:
                continue
            text = naive.random_text(rng, L)
            km, res = orc.spaced(naive.longseq_words(text, src), L, src, dst, K, J)
            assert res.status == 0
            assert rows(km) == naive.spaced(text, K, J, dst)
            assert len(km) == (0 if L < K else (L - K) // J + 1)  # SpacedKmers.jl:38-42


def test_spaced_error_semantics(orc):
    """J < K inspects every symbol up to the end of the last kmer; J >= K never looks at the
    gaps (SpacedKmers.jl:133-137, construction_utils.jl:213-214; test/runtests.jl:868-869)."""
    rng = np.random.default_rng(17)
    for K, J in [(3, 2), (21, 3), (3, 4), (5, 9), (31, 31)]:
        for L in (K, K + J, 200):
            text = naive.random_text(rng, L, p_amb=0.02)
            n = 0 if L < K else (L - K) // J + 1
            if J < K:
                inspected = range(1, (n - 1) * J + K + 1) if n else []
            else:
                inspected = [s * J + t + 1 for s in range(n) for t in range(K)]
            bad = naive.first_ambiguous(text, inspected)
            km, res = orc.spaced(naive.longseq_words(text, 4), L, 4, 2, K, J)
            if bad is None:
                assert res.status == 0 and rows(km) == naive.spaced(text, K, J, 2)
            else:
                assert res.status == 1 and res.err_pos == bad
                assert res.err_enc == naive.DNA4[text[bad - 1]]


@pytest.mark.parametrize("bps", [2, 4])
def test_transforms_random(orc, bps):
    rng = np.random.default_rng(19 + bps)
    for K in KS:
        if orc.nwords(K, bps) > 8:
            continue
        for _ in range(5):
            text = naive.random_text(rng, K, p_amb=0.2 if bps == 4 else 0.0)
            w = naive.kmer_words(text, bps)
            rc = naive.kmer_words(naive.revcomp_text(text), bps)
            assert orc.reverse(w, K, bps) == naive.kmer_words(text[::-1], bps)
            assert orc.complement(w, K, bps) == naive.kmer_words(naive.revcomp_text(text)[::-1], bps)
            assert orc.reverse_complement(w, K, bps) == rc
            assert orc.canonical_kmer(w, K, bps) == min(w, rc)
            assert orc.iscanonical(w, K, bps) == (w <= rc)
            assert orc.fx_hash(w, 12345) == naive.fx_hash(w, 12345)


def test_tuple_shift_quirks(orc):
    """left_shift/right_shift mask the count to 6 bits (tuple_bitflipping.jl:3-9): a zero-bit
    rightshift_carry must be the identity (used by reverse() when bits_unused == 0)."""
    import ctypes as C
    a = np.array([0x0123456789ABCDEF, 0xFEDCBA9876543210], dtype=np.uint64)
    b = a.copy()
    carry = orc.lib.orc_rightshift_carry(b.ctypes.data_as(C.POINTER(C.c_uint64)), 2, 0, 0)
    assert carry == 0 and list(b) == list(a)
    assert orc.lib.orc_get_mask(32, 2) == 2**64 - 1  # kmer.jl:603-605 with bits_unused == 0
    assert orc.lib.orc_get_mask(31, 2) == 2**62 - 1
    assert orc.lib.orc_get_mask(63, 2) == 2**62 - 1
    assert orc.lib.orc_get_mask(21, 4) == 2**20 - 1


def test_synth_generator_matches_numpy_model(orc):
    """The build's own generator: same base sequence in 2-bit and 4-bit form."""
    seed = 0x9E3779B97F4A7C15 ^ 2
    w2 = orc.synth_words(seed, 0, 8, 2)
    w4 = orc.synth_words(seed, 0, 16, 4)
    codes2 = [(int(w2[b // 32]) >> (2 * (b % 32))) & 3 for b in range(256)]
    nibs = [(int(w4[b // 16]) >> (4 * (b % 16))) & 15 for b in range(256)]
    assert nibs == [1 << c for c in codes2]
    # windowed generation is position independent
    assert list(orc.synth_words(seed, 5, 7, 4)) == list(w4[5:12])
    # ambiguity injection only replaces bases by N
    wa = orc.synth_words(seed, 0, 4096, 4, ambig_per_65536=2621)
    w0 = orc.synth_words(seed, 0, 4096, 4)
    na = 0
    for x, y in zip(wa.tolist(), w0.tolist()):
        for j in range(16):
            a, b = (x >> 4 * j) & 15, (y >> 4 * j) & 15
            assert a == b or a == 15
            na += a == 15
    assert 0.03 < na / (4096 * 16) < 0.05


def test_minimizers_definition(orc):
    """mode 1 = the kmer with the smallest fx_hash among W consecutive kmers (leftmost on ties);
    mode 0 = the reference's published example, checked against a literal Python transcription of
    docs/src/replacements.md:33-51 built on the naive packers."""
    rng = np.random.default_rng(31)
    for K, W, stride in [(5, 9, 1), (8, 20, 20), (31, 10, 7), (33, 5, 3), (4, 1, 2)]:
        for src in (2, 4):
            L = 600
            text = naive.random_text(rng, L)
            seq = naive.longseq_words(text, src)
            fw = naive.fw_kmers(text, K, 2)
            hs = [naive.fx_hash(w) for w in fw]
            n = (L - (K + W - 1)) // stride + 1
            got1, res = orc.minimizers(seq, L, src, 2, K, W, stride, mode=1)
            assert res.status == 0 and len(got1) == n
            exp1 = []
            for j in range(n):
                i = j * stride
                best = min(range(i, i + W), key=lambda t: (hs[t], t))
                exp1.append(fw[best])
            assert rows(got1) == exp1
            got0, res = orc.minimizers(seq, L, src, 2, K, W, stride, mode=0)
            exp0 = []
            mask = (1 << (2 * K)) - 1
            for j in range(n):
                i = j * stride
                v = int.from_bytes(b"", "big")
                kmer = 0
                for w in fw[i]:
                    kmer = (kmer << 64) | w
                nwords = len(fw[i])
                split = lambda x: tuple((x >> (64 * (nwords - 1 - t))) & (2**64 - 1) for t in range(nwords))
                h = naive.fx_hash(split(kmer))
                for off in range(W - 1):
                    code = naive.DNA2[text[i + K + off]]
                    nk = ((kmer << 2) | code) & mask   # shifted into the CURRENT MINIMUM, as published
                    nh = naive.fx_hash(split(nk))
                    if nh < h:
                        h, kmer = nh, nk
                exp0.append(split(kmer))
            assert rows(got0) == exp0


def test_oracle_at_many_words_matches_the_naive_slicer(orc):
    """The oracle's width limit is 64 words (ORC_MAX_N): FwRvIterator, CanonicalKmers + fx_hash, SpacedKmers and UnambiguousKmers of
    9 to 64 words against the independent big-integer slicer of tests/naive.py (the layout rule only, no shifting)."""
    import naive
    rng = np.random.default_rng(77)
    for src, dst, K in ((2, 2, 289), (4, 2, 700), (4, 2, 2048), (4, 4, 300), (2, 4, 1024), (8, 2, 1000)):
        L = K + 57
        text = naive.random_text(rng, L, p_amb=0.02 if dst == 4 and src != 2 else 0.0)
        words = naive.ascii_words(text) if src == 8 else naive.longseq_words(text, src)
        f, r, res = orc.fwrv(words, L, src, dst, K)
        assert res.status == 0
        pairs = naive.fwrv(text, K, dst)
        assert [tuple(int(x) for x in a) for a in f] == [tuple(a) for a, _ in pairs]
        assert [tuple(int(x) for x in a) for a in r] == [tuple(b) for _, b in pairs]
        k, h, _ = orc.canonical(words, L, src, dst, K, seed=9)
        canon = naive.canonical(text, K, dst)
        assert [tuple(int(x) for x in a) for a in k] == [tuple(c) for c in canon]
        assert h.tolist() == [naive.fx_hash(list(c), 9) for c in canon]
        s, _ = orc.spaced(words, L, src, dst, K, 7)
        assert [tuple(int(x) for x in a) for a in s] == [tuple(a) for a in naive.spaced(text, K, 7, dst)]
    text = "".join("N" if rng.random() < 0.002 else c for c in naive.random_text(rng, 6000))
    words = naive.longseq_words(text, 4)
    for K in (300, 1500):
        uk, us, _ = orc.unambiguous(words, len(text), 4, K)
        want = naive.unambiguous(text, K)
        assert [tuple(int(x) for x in a) for a in uk] == [tuple(a) for a, _ in want] and us.tolist() == [i for _, i in want]
