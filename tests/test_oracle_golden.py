"""Pins the CPU oracle against every known-answer vector the reference's own tests
hold for the hot path (tests/golden/kats.json)."""
import numpy as np
import pytest

import naive

BPS = {"dna2": 2, "rna2": 2, "dna4": 4, "rna4": 4, "aa": 8}


def words_of(arr_row):
    return tuple(int(x) for x in arr_row)


def test_g1_fx_hash(orc, kats):
    # test/runtests.jl:903-910 -- absolute bit patterns
    for c in kats["G1_fx_hash"]["cases"]:
        w = naive.kmer_words(c["text"], c["alphabet"])
        assert orc.fx_hash(w) == int(c["hash"], 16), c
        assert naive.fx_hash(w) == int(c["hash"], 16), c
    # SURVEY section 2.1 cross-check of the layout model: TAGCTAG -> 0x3272, UGAUGCA -> 0x38e4
    assert naive.kmer_words("TAGCTAG", "dna2") == (0x3272,)
    assert naive.kmer_words("UGAUGCA", "rna2") == (0x38E4,)


def test_g2_as_integer(orc, kats):
    # src/kmer.jl:288-298
    for c in kats["G2_as_integer"]["cases"]:
        bps = BPS[c["alphabet"]]
        w = naive.kmer_words(c["text"], c["alphabet"])
        val, bits = orc.as_integer(w, len(c["text"]), bps)
        assert val == int(c["value"], 16), c
        assert bits == c["bits"], c
        assert orc.from_integer(val, len(c["text"]), bps) == w


def test_g2b_from_integer(orc, kats):
    c0, c1 = kats["G2b_from_integer"]["cases"]
    assert orc.from_integer(int(c0["value"], 16), c0["K"], 2) == naive.kmer_words(c0["text"], "dna2")
    w = naive.kmer_words(c1["roundtrip_text"], "dna2")
    val, bits = orc.as_integer(w, c1["K"], 2)
    assert bits == c1["as_integer_bits"]
    assert orc.from_integer(val, c1["K"], 2) == w


def test_g3_fwrv(orc, kats):
    g = kats["G3_fwrv"]
    seq = naive.longseq_words(g["seq"], "dna4")
    fw, rv, res = orc.fwrv(seq, len(g["seq"]), 4, 4, g["K"])
    assert res.status == 0
    got = [(words_of(a), words_of(b)) for a, b in zip(fw, rv)]
    exp = [(naive.kmer_words(a, "dna4"), naive.kmer_words(b, "dna4")) for a, b in g["pairs"]]
    assert got == exp
    # same from a 2-bit source (TwoToFour)
    fw, rv, res = orc.fwrv(naive.longseq_words(g["seq"], "dna2"), len(g["seq"]), 2, 4, g["K"])
    assert [(words_of(a), words_of(b)) for a, b in zip(fw, rv)] == exp


def test_g4_canonical(orc, kats):
    g = kats["G4_canonical"]
    for src in (2, 4):
        seq = naive.longseq_words(g["seq"], src)
        km, hs, res = orc.canonical(seq, len(g["seq"]), src, 2, g["K"])
        assert res.status == 0
        assert [words_of(a) for a in km] == [naive.kmer_words(t, "rna2") for t in g["kmers"]]
        assert [int(h) for h in hs] == [naive.fx_hash(naive.kmer_words(t, "rna2")) for t in g["kmers"]]


def test_g5_fw(orc, kats):
    for c in kats["G5_fw"]["cases"]:
        for src in (2, 4):
            seq = naive.longseq_words(c["seq"], src)
            km, res = orc.fw_kmers(seq, len(c["seq"]), src, 2, c["K"])
            assert res.status == 0
            assert [words_of(a) for a in km] == [naive.kmer_words(t, "dna2") for t in c["kmers"]]


def test_g6_unambiguous(orc, kats):
    g = kats["G6_unambiguous"]
    seq = naive.longseq_words(g["seq"], 4)
    km, st, res = orc.unambiguous(seq, len(g["seq"]), 4, g["K"])
    assert res.status == 0
    got = [(words_of(a), int(s)) for a, s in zip(km, st)]
    assert got == [(naive.kmer_words(t, "rna2"), i) for t, i in g["items"]]


def test_g7_spaced(orc, kats):
    for c in kats["G7_spaced"]["cases"]:
        for src in (2, 4):
            seq = naive.longseq_words(c["seq"], src)
            km, res = orc.spaced(seq, len(c["seq"]), src, 2, c["K"], c["J"])
            assert res.status == 0
            assert [words_of(a) for a in km] == [naive.kmer_words(t, "dna2") for t in c["kmers"]]


def test_g8_construction_utils(orc, kats):
    g = kats["G8_construction_utils"]
    c = g["unsafe_extract"]  # reference uses an ASCII source; same symbols from 2/4-bit sources
    for src in (2, 4):
        w, res = orc.unsafe_extract(naive.longseq_words(c["seq"], src), src, 2, c["K"], c["from"])
        assert res.status == 0 and w == naive.kmer_words(c["kmer"], "dna2")
    c = g["shift_encoding"]
    assert orc.shift_encoding(naive.kmer_words(c["kmer"], "dna4"), 4, 4, c["enc"]) == \
        naive.kmer_words(c["result"], "dna4")
    c = g["unsafe_shift_from"]
    w, res = orc.unsafe_shift_from(naive.longseq_words(c["seq"], 4), 4, 2, 4, c["from"], c["S"],
                                   naive.kmer_words(c["kmer"], "dna2"))
    assert res.status == 0 and w == naive.kmer_words(c["result"], "dna2")


def test_g9_construction_tests(orc, kats):
    g = kats["G9_construction_tests"]
    text = g["seq"]
    for c in g["unsafe_extract"]:
        sb, db = BPS[c["src"]], BPS[c["dst"]]
        w, res = orc.unsafe_extract(naive.longseq_words(text, sb), sb, db, c["K"], c["from"])
        assert res.status == 0
        assert w == naive.kmer_words(text[c["from"] - 1:c["from"] - 1 + c["K"]], c["dst"]), c
    for c in g["unsafe_shift_from"]:
        sb, db = BPS[c["src"]], BPS[c["dst"]]
        K = len(c["kmer"])
        w, res = orc.unsafe_shift_from(naive.longseq_words(text, sb), sb, db, K, c["from"], c["S"],
                                       naive.kmer_words(c["kmer"], c["dst"]))
        assert res.status == 0
        assert w == naive.kmer_words(c["result"], c["dst"]), c


def test_g10_iscanonical(orc, kats):
    g = kats["G10_iscanonical"]
    for t in g["true"]:
        assert orc.iscanonical(naive.kmer_words(t, "dna2"), len(t), 2), t
        assert orc.iscanonical(naive.kmer_words(t, "dna4"), len(t), 4), t
    for t in g["false"]:
        assert not orc.iscanonical(naive.kmer_words(t, "dna2"), len(t), 2), t
        assert not orc.iscanonical(naive.kmer_words(t, "dna4"), len(t), 4), t


def test_g11_shift(orc, kats):
    g = kats["G11_shift"]
    for bps, tab in ((2, naive.DNA2), (4, naive.DNA4)):
        c = g["shift"]
        assert orc.shift_encoding(naive.kmer_words(c["kmer"], bps), 4, bps, tab[c["symbol"]]) == \
            naive.kmer_words(c["result"], bps)
        c = g["shift_first"]
        assert orc.shift_first_encoding(naive.kmer_words(c["kmer"], bps), 4, bps, tab[c["symbol"]]) == \
            naive.kmer_words(c["result"], bps)


def test_g12_errors(orc, kats):
    for c in kats["G12_errors"]["cases"]:
        seq = naive.longseq_words(c["seq"], 4)
        if c["iter"] == "fw":
            _, res = orc.fw_kmers(seq, len(c["seq"]), 4, 2, c["K"])
            _, _, res2 = orc.fwrv(seq, len(c["seq"]), 4, 2, c["K"])
            assert (res2.status, res2.err_pos, res2.err_enc) == (res.status, res.err_pos, res.err_enc)
        else:
            _, res = orc.spaced(seq, len(c["seq"]), 4, 2, c["K"], c["J"])
        assert res.status == 1, c
        assert res.err_pos == c["err_pos"], c
        assert res.err_enc == naive.DNA4[c["err_symbol"]], c


def test_g13_gc_count(orc, kats):
    # test/runtests.jl:1021-1027 (src/counting.jl:1-8)
    for text, n in kats["G13_gc_count"]["cases"]:
        assert orc.n_gc(naive.kmer_words(text, 2)) == n, text


def test_g14_reference_property_sequences(orc, kats):
    """The fixed sequences of the reference's differential tests, checked against the
    naive slicer (test/runtests.jl:674-690, :697-711, :739-761, :774-787, :805-847, :850-867)."""
    g = kats["G14_property_seqs"]
    for text in g["fw_two_bit"] + g["fw_four_bit"]:
        for src in (2, 4):
            for dst in (2, 4):
                km, res = orc.fw_kmers(naive.longseq_words(text, src), len(text), src, dst, 3)
                assert res.status == 0
                assert [words_of(a) for a in km] == naive.fw_kmers(text, 3, dst)
    for text in g["four_to_two"]:
        km, res = orc.fw_kmers(naive.longseq_words(text, 4), len(text), 4, 2, 4)
        assert [words_of(a) for a in km] == naive.fw_kmers(text, 4, 2)
    for text in g["fwrv"]:
        for src in (2, 4):
            for dst in (2, 4):
                fw, rv, res = orc.fwrv(naive.longseq_words(text, src), len(text), src, dst, 4)
                assert res.status == 0
                assert [(words_of(a), words_of(b)) for a, b in zip(fw, rv)] == naive.fwrv(text, 4, dst)
    for text in g["canonical"]:
        for src in (2, 4):
            for dst in (2, 4):
                km, _, res = orc.canonical(naive.longseq_words(text, src), len(text), src, dst, 5)
                assert res.status == 0
                assert [words_of(a) for a in km] == naive.canonical(text, 5, dst)
    for text in g["unambiguous"]:
        for K in (3, 4):
            km, st, res = orc.unambiguous(naive.longseq_words(text, 4), len(text), 4, K)
            assert [(words_of(a), int(s)) for a, s in zip(km, st)] == naive.unambiguous(text, K)
    for text in g["unambiguous_two_bit"]:
        km, st, res = orc.unambiguous(naive.longseq_words(text, 2), len(text), 2, 4)
        assert [(words_of(a), int(s)) for a, s in zip(km, st)] == naive.unambiguous(text, 4)
    for c in g["spaced"]:
        text, dst = c["seq"], BPS[c["dst"]]
        srcs = (4,) if not naive.is_certain(text) else (2, 4)
        for K, J in c["KJ"]:
            for src in srcs:
                km, res = orc.spaced(naive.longseq_words(text, src), len(text), src, dst, K, J)
                assert res.status == 0
                assert [words_of(a) for a in km] == naive.spaced(text, K, J, dst)


@pytest.mark.parametrize("bps", [2, 4])
def test_longseq_kmer_roundtrip(orc, bps):
    # src/construction.jl:213-219 and :289-324 are inverse maps; both agree with the naive packers
    rng = np.random.default_rng(5)
    for K in (1, 5, 16, 31, 32, 33, 47, 63, 64, 65, 96):
        if orc.nwords(K, bps) > 8:
            continue
        text = naive.random_text(rng, K)
        ls = naive.longseq_words(text, bps)
        km = orc.kmer_from_longseq(ls, K, K, bps)
        assert km == naive.kmer_words(text, bps)
        back = orc.longseq_from_kmer(km, K, bps)
        assert list(back) == list(ls[:len(back)])


def test_synth_10k_fixture(orc):
    """tests/golden/synth_10k.json: the committed generator + end-to-end pin.
    The first and last 16 canonical kmers are also re-derived with the naive big-int slicer from the
    decoded text, so the fixture is not only the oracle agreeing with itself."""
    import json
    import os
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_10k.json")))
    L, seed = fx["n_bases"], int(fx["seed"], 16)
    for c in fx["cases"]:
        bits = c["src_bits"]
        if c["iter"] == "canonical":
            n_words = (L * bits + 63) // 64
            words = orc.synth_words(seed, 0, n_words + 1, bits)
            assert [f"0x{int(x):016x}" for x in words[:4]] == c["source_words_first4"]
            assert int(np.bitwise_xor.reduce(words[:n_words])) == int(c["source_xor"], 16)
            km, hs, res = orc.canonical(words, L, bits, 2, c["K"])
            assert res.status == 0 and len(km) == c["n"]
            assert [int(np.bitwise_xor.reduce(km[:, j])) for j in range(km.shape[1])] == [int(x, 16) for x in c["kmer_xor"]]
            assert int(np.bitwise_xor.reduce(hs)) == int(c["hash_xor"], 16)
            # decode the LongSequence words symbol by symbol (LE packing) and slice naively
            sym = "ACGT"
            if bits == 2:
                text = "".join(sym[(int(words[i // 32]) >> (2 * (i % 32))) & 3] for i in range(L))
            else:
                text = "".join(sym[((int(words[i // 16]) >> (4 * (i % 16))) & 15).bit_length() - 1] for i in range(L))
            N = km.shape[1]
            exp = naive.canonical(text[:c["K"] + 15], c["K"], 2) + naive.canonical(text[-(c["K"] + 15):], c["K"], 2)
            flat = [int(x, 16) for x in c["first16"] + c["last16"]]
            assert [tuple(flat[i * N:(i + 1) * N]) for i in range(32)] == exp
            assert [int(x, 16) for x in c["first16_hashes"] + c["last16_hashes"]] == [naive.fx_hash(w) for w in exp]
        else:
            words = orc.synth_words(int(c["seed"], 16), 0, L // 16 + 1, 4, c["ambig_per_65536"])
            km, st, res = orc.unambiguous(words, L, 4, c["K"])
            assert len(km) == c["n"] and int(st.sum()) == c["start_sum"]
            assert [int(x) for x in st[:16]] == c["first16_starts"]
            assert int(np.bitwise_xor.reduce(km[:, 0])) == int(c["kmer_xor"], 16)
            _, sres = orc.spaced(words, L, 4, 2, c["K"], 3)
            assert sres.status == 1 and (sres.err_pos, sres.err_enc) == (c["strict_spaced_error"]["pos"], c["strict_spaced_error"]["enc"])


def test_julia_comparison_tool_self_test():
    """tests/golden/compare_with_julia.py (the tool that pins the oracle on vectors from the real
    reference when Julia is available) runs, and oracle == naive model on the vectors it covers."""
    import os
    import subprocess
    import sys
    tool = os.path.join(os.path.dirname(__file__), "golden", "compare_with_julia.py")
    r = subprocess.run([sys.executable, tool, "--self-test"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all vectors agree" in r.stdout, r.stdout + r.stderr
