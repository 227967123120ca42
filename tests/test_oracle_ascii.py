"""AsciiEncode sources in the oracle (String / Vector{UInt8}: FwKmers.jl:69-78,117-129;
CanonicalKmers.jl:69-79,146-174; UnambiguousKmers.jl:109-132; SpacedKmers.jl:109-119;
construction_utils.jl:71-88,220-236), pinned on the reference's own ASCII tests
(test/runtests.jl:713-725, :774-801, :822-847, :850-870; docstrings)."""
import numpy as np

import naive

DNA_A, RNA_A = 8, 9  # ORC_SRC_ASCII_DNA / ORC_SRC_ASCII_RNA


def rows(a):
    return [tuple(int(x) for x in r) for r in a]


def test_docstring_and_reference_ascii_cases(orc):
    # construction_utils.jl:20-24: unsafe_extract(AsciiEncode(), DNAKmer{4,1}, b"TAGCTAGA", 2) == AGCT
    w, res = orc.unsafe_extract(naive.ascii_words("TAGCTAGA"), DNA_A, 2, 4, 2)
    assert res.status == 0 and w == naive.kmer_words("AGCT", 2)
    # FwKmers.jl:14-22 uses a String source
    km, res = orc.fw_kmers(naive.ascii_words("AGCGTATA"), 8, DNA_A, 2, 3)
    assert rows(km) == naive.fw_kmers("AGCGTATA", 3, 2)
    # CanonicalKmers.jl:14-18 FwRvIterator{DNAAlphabet{4},3}("AGCGT")
    fw, rv, res = orc.fwrv(naive.ascii_words("AGCGT"), 5, DNA_A, 4, 3)
    assert list(zip(rows(fw), rows(rv))) == naive.fwrv("AGCGT", 3, 4)
    # CanonicalKmers.jl:20-22: FwRvIterator{DNAAlphabet{2},3}("AGNGT") -> cannot encode 0x4e (Char 'N')
    _, _, res = orc.fwrv(naive.ascii_words("AGNGT"), 5, DNA_A, 2, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, 0x4E)
    # CanonicalKmers.jl:192-196 CanonicalRNAMers{3}("AGCGA")
    ck, _, res = orc.canonical(naive.ascii_words("AGCGA"), 5, RNA_A, 2, 3)
    assert rows(ck) == [naive.kmer_words(t, "rna2") for t in ("AGC", "CGC", "CGA")]
    # SpacedKmers.jl:16-20 and each_codon(DNA, "TGACGATCGAC") (:70-75)
    km, _ = orc.spaced(naive.ascii_words("AGCGTATA"), 8, DNA_A, 2, 3, 2)
    assert rows(km) == [naive.kmer_words(t, 2) for t in ("AGC", "CGT", "TAT")]
    km, _ = orc.spaced(naive.ascii_words("TGACGATCGAC"), 11, DNA_A, 2, 3, 3)
    assert rows(km) == [naive.kmer_words(t, 2) for t in ("TGA", "CGA", "TCG")]
    # test/runtests.jl:868-869 SpacedDNAMers{3,4}("TAGAWWWW") throws
    _, res = orc.spaced(naive.ascii_words("TAGAWWWW"), 8, DNA_A, 2, 3, 4)
    assert (res.status, res.err_pos, res.err_enc) == (1, 5, ord("W"))
    # test/runtests.jl:722-724, :845-846: bad byte 'P'
    s = "TAGTCGTAGPATGC"
    _, res = orc.fw_kmers(naive.ascii_words(s), len(s), DNA_A, 2, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 10, ord("P"))
    _, _, res = orc.unambiguous(naive.ascii_words(s), len(s), DNA_A, 3)
    assert (res.status, res.err_pos, res.err_enc) == (1, 10, ord("P"))


def test_mixed_case_iupac_string(orc):
    # test/runtests.jl:713-720: "TaghWS-TGnADbkWWMSTV" as FwKmers{DNAAlphabet{4},4}
    s = "TaghWS-TGnADbkWWMSTV"
    km, res = orc.fw_kmers(naive.ascii_words(s), len(s), DNA_A, 4, 4)
    assert res.status == 0 and rows(km) == naive.fw_kmers(s, 4, 4)
    # same symbols from the 4-bit LongSequence
    km2, _ = orc.fw_kmers(naive.longseq_words(s, 4), len(s), 4, 4, 4)
    assert np.array_equal(km, km2)


def test_dna_rna_validity(orc):
    # U is not a DNA symbol, T is not an RNA symbol (BioSequences.ascii_encode)
    _, res = orc.fw_kmers(naive.ascii_words("ACGU"), 4, DNA_A, 2, 2)
    assert (res.status, res.err_pos, res.err_enc) == (1, 4, ord("U"))
    _, res = orc.fw_kmers(naive.ascii_words("ACGT"), 4, RNA_A, 2, 2)
    assert (res.status, res.err_pos, res.err_enc) == (1, 4, ord("T"))
    km, res = orc.fw_kmers(naive.ascii_words("acgu"), 4, RNA_A, 4, 2)
    assert res.status == 0 and rows(km) == naive.fw_kmers("ACGU", 2, 4)
    # the skipping LUT of UnambiguousKmers accepts both T and U (common.jl:22-32)
    km, st, res = orc.unambiguous(naive.ascii_words("ACGUTacgut"), 10, DNA_A, 3)
    assert res.status == 0 and len(km) == 8
    # non-ASCII bytes are invalid everywhere
    _, res = orc.fw_kmers(naive.ascii_words(b"AC\xc3\xa9GT"), 6, DNA_A, 4, 2)
    assert (res.status, res.err_pos, res.err_enc) == (1, 3, 0xC3)


def test_ascii_equals_longsequence_random(orc):
    rng = np.random.default_rng(77)
    for K in (1, 3, 16, 31, 33, 64):
        for L in (K - 1, K, 200):
            if L < 0:
                continue
            text = naive.random_text(rng, L)
            mixed = "".join(c.lower() if rng.random() < 0.3 else c for c in text)
            for dst in (2, 4):
                if orc.nwords(K, dst) > 8:
                    continue
                a_fw, a_rv, ra = orc.fwrv(naive.ascii_words(mixed), L, DNA_A, dst, K)
                b_fw, b_rv, rb = orc.fwrv(naive.longseq_words(text, 4), L, 4, dst, K)
                assert ra.status == rb.status == 0
                assert np.array_equal(a_fw, b_fw) and np.array_equal(a_rv, b_rv)
                ck, hs, _ = orc.canonical(naive.ascii_words(mixed), L, DNA_A, dst, K, seed=9)
                ck2, hs2, _ = orc.canonical(naive.longseq_words(text, 2), L, 2, dst, K, seed=9)
                assert np.array_equal(ck, ck2) and np.array_equal(hs, hs2)
            amb = naive.random_text(rng, L, p_amb=0.1)
            km, st, res = orc.unambiguous(naive.ascii_words(amb.lower()), L, DNA_A, K)
            assert res.status == 0
            assert list(zip(rows(km), [int(x) for x in st])) == naive.unambiguous(amb, K)


def test_computed_ascii_tables_equal_the_letter_list_tables(tmp_path):
    """kmers.jl_amd/csrc/ascii_tables.hpp: ascii_entry() (what the kernels use: table entries computed from two
    immediates, no memory) == the tables built from the letter lists, for all 5 x 256 entries (plain C++)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "ascii_entry_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", str(exe), os.path.join(root, "tests", "c", "ascii_entry_check.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "agrees" in r.stdout, r.stdout
