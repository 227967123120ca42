"""GenericRecoding sources in the oracle (src/construction.jl:90-98: collections of nucleotide symbols that are neither a
BioSequence nor text, e.g. Vector{DNA}; FwKmers.jl:80-86, CanonicalKmers.jl:81-91, construction_utils.jl:90-103,
:161-172): one BioSymbols value per byte.  The reference's own tests hold such sources equal to the BioSequence of the same
symbols (test/runtests.jl:674-690 iterate `collect(seq)`-like generic sources), which is the differential used here: a
symbol vector must give exactly what the LongSequence{DNAAlphabet{4}} of the same symbols gives -- kmers and EncodeErrors."""
import json
import os

import numpy as np

import naive

SYMBOLS = 10  # ORC_SRC_SYMBOLS
ENC4 = {c: i for i, c in enumerate("-ACMGRSVTWYHKDBN")}


def symbol_words(text):
    return naive.ascii_words(bytes(ENC4[c] for c in text.upper()))


def test_symbol_vectors_equal_the_four_bit_sequence_of_the_same_symbols(orc):
    rng = np.random.default_rng(2026)
    for dst in (2, 4):
        for K in (1, 3, 21, 31, 33, 64):
            for L in (K, K + 5, 700):
                for p_amb in (0.0, 0.02):
                    text = naive.random_text(rng, L, p_amb=p_amb)
                    sv, ls = symbol_words(text), naive.longseq_words(text, 4)
                    a, ra = orc.fw_kmers(sv, L, SYMBOLS, dst, K)
                    b, rb = orc.fw_kmers(ls, L, 4, dst, K)
                    assert (ra.status, ra.err_pos, ra.err_enc) == (rb.status, rb.err_pos, rb.err_enc)
                    if ra.status == 0:
                        assert np.array_equal(a, b)
                    fa, va, ra = orc.fwrv(sv, L, SYMBOLS, dst, K)
                    fb, vb, rb = orc.fwrv(ls, L, 4, dst, K)
                    assert (ra.status, ra.err_pos, ra.err_enc) == (rb.status, rb.err_pos, rb.err_enc)
                    if ra.status == 0:
                        assert np.array_equal(fa, fb) and np.array_equal(va, vb)
                    ca, ha, ra = orc.canonical(sv, L, SYMBOLS, dst, K, seed=11)
                    cb, hb, rb = orc.canonical(ls, L, 4, dst, K, seed=11)
                    assert ra.status == rb.status and (ra.status or (np.array_equal(ca, cb) and np.array_equal(ha, hb)))
                    for J in (1, 3, K + 2):
                        sa, ra = orc.spaced(sv, L, SYMBOLS, dst, K, J)
                        sb, rb = orc.spaced(ls, L, 4, dst, K, J)
                        assert (ra.status, ra.err_pos, ra.err_enc) == (rb.status, rb.err_pos, rb.err_enc)
                        if ra.status == 0:
                            assert np.array_equal(sa, sb)


def test_known_cases(orc):
    # FwKmers.jl:14-22 with the symbols as a vector: FwDNAMers{3}(collect(dna"AGCGTATA"))
    km, res = orc.fw_kmers(symbol_words("AGCGTATA"), 8, SYMBOLS, 2, 3)
    assert res.status == 0 and [tuple(int(x) for x in r) for r in km] == naive.fw_kmers("AGCGTATA", 3, 2)
    # a gap or an ambiguity code cannot be encoded in a 2-bit alphabet (BioSequences.encode): EncodeError(A, symbol)
    _, res = orc.fw_kmers(symbol_words("AG-GT"), 5, SYMBOLS, 2, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, 0)
    _, res = orc.fw_kmers(symbol_words("AGWGT"), 5, SYMBOLS, 2, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, ENC4["W"])
    # ... but both are symbols of the 4-bit alphabets
    km, res = orc.fw_kmers(symbol_words("AG-WT"), 5, SYMBOLS, 4, 5)
    assert res.status == 0 and tuple(int(x) for x in km[0]) == tuple(naive.kmer_words("AG-WT", 4))
    # a byte that is no nucleotide value at all
    _, res = orc.fw_kmers(naive.ascii_words(bytes([1, 2, 0x41, 4])), 4, SYMBOLS, 4, 2)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, 0x41)


def test_unambiguous_over_symbols_is_the_generic_method(orc):
    """UnambiguousKmers over a collection of symbols takes the method of `::RecodingScheme` (UnambiguousKmers.jl:88-106): an
    ambiguous symbol is skipped, a certain one shifted in -- the elements of the four-bit sequence of the same symbols, as the
    reference's own test compares them (test/runtests.jl:826-840: LongSequence{GenericNucAlphabet} against the naive filter)
    -- and the gap, which is neither, fails in `shift`'s encode where the four-bit method skips it."""
    rng = np.random.default_rng(17)
    kats = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kats.json")))
    texts = [t.replace("-", "N") for t in kats["G14_property_seqs"]["unambiguous"]] + ["TGATCGTAGATGWATGTC"]  # (runtests.jl:836)
    texts += ["".join(rng.choice(list("ACGTACGTACGTNWSK"), int(L))) for L in (1, 5, 40, 333, 2000)]
    for text in texts:
        for K in (1, 3, 4, 7, 31, 33):
            ka, sa, ra = orc.unambiguous(symbol_words(text), len(text), SYMBOLS, K)
            kb, sb, rb = orc.unambiguous(naive.longseq_words(text, 4), len(text), 4, K)
            assert ra.status == rb.status == 0 and np.array_equal(ka, kb) and np.array_equal(sa, sb), (text[:20], K)
            exp = naive.unambiguous(text, K)
            assert [tuple(int(x) for x in r) for r in ka] == [k_ for k_, _ in exp] and list(sa) == [i for _, i in exp]
    # the gap: skipped in a four-bit sequence (count_ones != 1, :140-146), an EncodeError here -- also in a sequence shorter than K
    k4, s4, r4 = orc.unambiguous(naive.longseq_words("ACG-TACGT", 4), 9, 4, 3)
    assert r4.status == 0 and list(s4) == [1, 5, 6, 7]
    _, _, res = orc.unambiguous(symbol_words("ACG-TACGT"), 9, SYMBOLS, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 4, 0)
    _, _, res = orc.unambiguous(symbol_words("A-"), 2, SYMBOLS, 5)
    assert (res.status, res.err_pos, res.err_enc) == (1, 2, 0)
    _, _, res = orc.unambiguous(naive.ascii_words(bytes([1, 2, 0x41, 4])), 4, SYMBOLS, 2)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, 0x41)
