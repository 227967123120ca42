"""The reference's own iterator tests, re-expressed against the host mirror (kmers_jl_amd), which
runs everything through libkmers_hip.so.  Mirrors test/runtests.jl:660-889 and the docstring
examples of src/iterators/*.jl.  Run with -m gpu."""
import re

import numpy as np
import pytest

import naive

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    return kmers_jl_amd


def texts(xs):
    return [str(x) for x in xs]


def test_docstring_examples(km, kats):
    # FwKmers.jl:14-22
    c = kats["G5_fw"]["cases"][0]
    it = km.FwDNAMers[3](km.LongDNA[4](c["seq"]))
    assert texts(km.collect(it)) == c["kmers"] and len(it) == 6
    assert it.eltype == ("Kmer", km.DNAAlphabet[2], 3, 1)
    # docs/src/iteration.md:19-21  only(FwDNAMers{3}(rna"UGU")) == mer"TGT"d
    assert km.collect(km.FwDNAMers[3](km.LongRNA[4]("UGU")))[0] == km.mer("TGT")
    # CanonicalKmers.jl:192-196
    g = kats["G4_canonical"]
    assert texts(km.collect(km.CanonicalRNAMers[3](km.LongDNA[4](g["seq"])))) == g["kmers"]
    # UnambiguousKmers.jl:20-26
    g = kats["G6_unambiguous"]
    got = km.collect(km.UnambiguousRNAMers[4](km.LongDNA[4](g["seq"])))
    assert [(str(k), i) for k, i in got] == [tuple(x) for x in g["items"]]
    # SpacedKmers.jl:16-20 and :70-75
    for c in kats["G7_spaced"]["cases"]:
        assert texts(km.collect(km.SpacedDNAMers[c["K"], c["J"]](km.LongDNA[4](c["seq"])))) == c["kmers"]
    assert texts(km.collect(km.each_codon(km.LongDNA[2]("TGACGATCGAC")))) == ["TGA", "CGA", "TCG"]


def test_smaller_than_k_and_aliases(km):
    # test/runtests.jl:668-672
    assert len(km.FwDNAMers[3](km.LongDNA[4]("TA"))) == 0
    assert km.collect(km.FwDNAMers[3](km.LongDNA[4]("TA"))).tolist() == []
    assert km.collect(km.CanonicalDNAMers[8](km.LongDNA[2](""))).tolist() == []
    assert isinstance(km.FwKmers[km.DNAAlphabet[2], 3](km.LongDNA[4]("TAGA")), km.FwKmers)
    with pytest.raises(km.KmersError, match="K must be at least 1"):
        km.FwDNAMers[0](km.LongDNA[4]("TAGA"))
    with pytest.raises(km.KmersError, match="J must be at least 1"):
        km.SpacedDNAMers[3, 0](km.LongDNA[4]("TAGA"))


def test_conversible_alphabets(km, kats):
    # test/runtests.jl:674-690: every source alphabet x every kmer alphabet
    g = kats["G14_property_seqs"]
    for text in g["fw_two_bit"] + g["fw_four_bit"]:
        for src in (km.LongDNA[2], km.LongDNA[4], km.LongRNA[2], km.LongRNA[4]):
            for alphabet in (km.DNAAlphabet[2], km.DNAAlphabet[4], km.RNAAlphabet[2], km.RNAAlphabet[4]):
                v1 = km.collect(km.FwKmers[alphabet, 3](src(text)))
                assert [k.data for k in v1] == naive.fw_kmers(text, 3, alphabet.bits)
    # docstring of FwRvIterator (CanonicalKmers.jl:14-18): 4-bit kmers
    g3 = kats["G3_fwrv"]
    got = km.collect(km.FwRvIterator[km.DNAAlphabet[4], 3](km.LongDNA[4](g3["seq"])))
    assert [(str(a), str(b)) for a, b in got] == [tuple(p) for p in g3["pairs"]]
    # 4-bit kmers keep ambiguous symbols (Copyable): test/runtests.jl:857, :866
    amb = "TAGCCWKMMNAGCTV"
    got = km.collect(km.SpacedKmers[km.RNAAlphabet[4], 2, 3](km.LongDNA[4](amb)))
    assert [k.data for k in got] == naive.spaced(amb, 2, 3, 4)


def test_ambiguous_sources_throw(km, kats):
    # test/runtests.jl:691-694; FwKmers.jl:24-25 "cannot encode D in RNAAlphabet{2}"
    for c in kats["G12_errors"]["cases"]:
        seq = km.LongDNA[4](c["seq"])
        if c["iter"] == "fw":
            its = [km.FwDNAMers[c["K"]](seq), km.CanonicalDNAMers[c["K"]](seq), km.FwRvDNAIterator[c["K"]](seq)]
        else:
            its = [km.SpacedDNAMers[c["K"], c["J"]](seq)]
        for it in its:
            with pytest.raises(km.EncodeError) as e:
                km.collect(it)
            assert e.value.position == c["err_pos"] and e.value.symbol == c["err_symbol"]
    with pytest.raises(km.EncodeError, match=re.escape("cannot encode D in RNAAlphabet{2}")):
        km.collect(km.FwRNAMers[3](km.LongRNA[4]("UGCDUGAVC")))


def test_iteration_yields_until_the_throw(km):
    """`for x in it` yields exactly what the reference yields before throwing (FwKmers.jl:104-115)."""
    text = "ACGTTGCAAC" + "W" + "ACGT"
    it = km.FwDNAMers[4](km.LongDNA[4](text))
    got = []
    with pytest.raises(km.EncodeError) as e:
        for k in it:
            got.append(str(k))
    assert e.value.position == 11
    assert got == [text[i:i + 4] for i in range(0, 7)]  # windows ending before symbol 11
    # spaced, J >= K: the gap symbol is never inspected, the next kmer's is
    it = km.SpacedDNAMers[3, 4](km.LongDNA[4]("TAGWTAGATAWA"))
    got = []
    with pytest.raises(km.EncodeError) as e:
        for k in it:
            got.append(str(k))
    assert got == ["TAG", "TAG"] and e.value.position == 11


def test_fwrv_and_canonical_vs_naive(km, kats):
    # test/runtests.jl:739-761, :774-787
    g = kats["G14_property_seqs"]
    for text in g["fwrv"]:
        for src in (km.LongDNA[2], km.LongDNA[4], km.LongRNA[2], km.LongRNA[4]):
            got = km.collect(km.FwRvDNAIterator[4](src(text)))
            assert [(a.data, b.data) for a, b in got] == naive.fwrv(text, 4, 2)
    for text in g["canonical"]:
        for src in (km.LongDNA[2], km.LongDNA[4]):
            got = km.collect(km.CanonicalDNAMers[5](src(text)))
            assert [k.data for k in got] == naive.canonical(text, 5, 2)
            # equivalent to calling canonical on each FwKmers element (CanonicalKmers.jl:179-181)
            fw = km.collect(km.FwDNAMers[5](src(text)))
            assert km.canonical(fw) == got


def test_unambiguous_vs_filter(km, kats):
    # test/runtests.jl:803-847
    g = kats["G14_property_seqs"]
    for text in g["unambiguous"]:
        for K in (3, 4):
            got = km.collect(km.UnambiguousDNAMers[K](km.LongDNA[4](text)))
            assert [(k.data, i) for k, i in got] == naive.unambiguous(text, K)
    s = km.LongDNA[2]("TATCGGATAGGCAAA")
    it = km.UnambiguousRNAMers[4](s)
    assert len(it) == 12  # HasLength for 2-bit sources (UnambiguousKmers.jl:34-37)
    assert [(k.data, i) for k, i in km.collect(it)] == naive.unambiguous(str(s), 4)
    with pytest.raises(TypeError):
        len(km.UnambiguousDNAMers[4](km.LongDNA[4]("ACGT")))


def test_spaced_vs_naive(km, kats):
    # test/runtests.jl:849-867 (2-bit kmer alphabets)
    for c in kats["G14_property_seqs"]["spaced"]:
        if c["dst"] != "dna2":
            continue
        for K, J in c["KJ"]:
            for src in (km.LongDNA[2], km.LongDNA[4]):
                it = km.SpacedDNAMers[K, J](src(c["seq"]))
                got = km.collect(it)
                assert [k.data for k in got] == naive.spaced(c["seq"], K, J, 2)
                assert len(it) == len(got)


def test_string_sources(km, kats):
    """String / byte sources (test/runtests.jl:713-725, :774-801, :836-847, :850-870)."""
    s = "TaghWS-TGnADbkWWMSTV"
    T = km.FwKmers[km.DNAAlphabet[4], 4]
    mers = km.collect(T(s))
    for source in (s, s.encode(), bytearray(s.encode()), np.frombuffer(s.encode(), dtype=np.uint8)):
        assert km.collect(T(source)) == mers
    assert mers == km.collect(T(km.LongDNA[4](s)))
    with pytest.raises(km.EncodeError):
        km.collect(km.FwDNAMers[3]("TAGTCGTAGPATGC"))
    with pytest.raises(km.EncodeError, match=re.escape("cannot encode 0x4e (Char 'N') in DNAAlphabet{2}")):
        km.collect(km.FwRvDNAIterator[3]("AGNGT"))                       # CanonicalKmers.jl:20-22
    with pytest.raises(km.EncodeError, match=re.escape("cannot encode 0x55 (Char 'U') in DNAAlphabet{2}")):
        km.collect(km.FwDNAMers[3]("UGU"))                               # docs/src/iteration.md:23-25: text is not converted between DNA and RNA
    assert km.collect(km.FwRNAMers[3]("UGU"))[0] == km.mer("UGU", "r")
    for text in kats["G14_property_seqs"]["canonical"]:
        got = km.collect(km.CanonicalDNAMers[5](text))
        assert [k.data for k in got] == naive.canonical(text, 5, 2)
    assert km.collect(km.CanonicalKmers[km.DNAAlphabet[2], 4]("TAGTGTCGATGATC")) == \
        km.collect(km.CanonicalDNAMers[4]("TAGTGTCGATGATC"))
    for text in kats["G14_property_seqs"]["unambiguous"]:
        got = km.collect(km.UnambiguousDNAMers[4](text))
        assert [(k.data, i) for k, i in got] == naive.unambiguous(text, 4)
    with pytest.raises(km.EncodeError):
        km.collect(km.UnambiguousDNAMers[3]("TAGTCGTAGPATGC"))
    with pytest.raises(km.EncodeError):
        km.collect(km.SpacedDNAMers[3, 4]("TAGAWWWW"))                   # test/runtests.jl:868-869
    assert texts(km.collect(km.each_codon(km.DNA, "TGACGATCGAC"))) == ["TGA", "CGA", "TCG"]  # SpacedKmers.jl:70-75
    assert texts(km.collect(km.each_codon(km.RNA, b"UAUGCUGAA"))) == ["UAU", "GCU", "GAA"]


def test_symbol_vector_sources(km):
    """A Vector{DNA} / Vector{RNA} source goes through GenericRecoding in the reference (src/construction.jl:90-98,
    FwKmers.jl:80-86): the same elements as the BioSequence of the same symbols, the same EncodeError."""
    text = "TAGCTGACCGTTAGGCATCGATCGGATCCGATAGCTAGCTAGGA"
    v = km.SymbolVector("DNA", text)
    assert str(v) == text and len(v) == len(text)
    for K in (3, 21, 33):
        assert km.collect(km.FwDNAMers[K](v)) == km.collect(km.FwDNAMers[K](km.LongDNA[4](text)))
        assert km.collect(km.CanonicalDNAMers[K](v)) == km.collect(km.CanonicalDNAMers[K](km.LongDNA[4](text)))
        assert km.collect(km.FwRNAMers[K](v)) == km.collect(km.FwRNAMers[K](km.LongDNA[4](text)))       # convert(RNA, ::DNA)
        assert km.collect(km.SpacedDNAMers[K, 2](v)) == km.collect(km.SpacedDNAMers[K, 2](km.LongDNA[4](text)))
    amb = km.SymbolVector("DNA", "TAGWC-GA")
    assert km.collect(km.FwKmers[km.DNAAlphabet[4], 3](amb)) == km.collect(km.FwKmers[km.DNAAlphabet[4], 3](km.LongDNA[4]("TAGWC-GA")))
    with pytest.raises(km.EncodeError, match=re.escape("cannot encode W in DNAAlphabet{2}")):
        km.collect(km.FwDNAMers[3](amb))
    got = []
    with pytest.raises(km.EncodeError):
        for x in km.FwDNAMers[3](amb):      # TAG is yielded, then AGW throws
            got.append(str(x))
    assert got == ["TAG"]
    # UnambiguousKmers over such a source is the reference's generic method (UnambiguousKmers.jl:88-106, runtests.jl:834-840:
    # LongSequence{GenericNucAlphabet}(dna"TGATCGTAGATGwATGTC")): the elements of the four-bit sequence of the same symbols ...
    for t in (text, "TGATCGTAGATGWATGTC", "TAGCTKAGAGGAGAACWSGCGAGA"):
        for K in (4, 7):
            assert km.collect(km.UnambiguousDNAMers[K](km.SymbolVector("DNA", t))) == km.collect(km.UnambiguousDNAMers[K](km.LongDNA[4](t)))
    # ... except for the gap: a four-bit sequence skips it, `shift` of a symbol cannot encode it
    assert [i for _, i in km.collect(km.UnambiguousDNAMers[3](km.LongDNA[4]("TAGWC-GAT")))] == [1, 7]
    with pytest.raises(km.EncodeError, match=re.escape("cannot encode - in DNAAlphabet{2}")):
        km.collect(km.UnambiguousDNAMers[3](km.SymbolVector("DNA", "TAGWC-GAT")))


def test_fx_hash_known_answers(km, kats):
    # test/runtests.jl:903-910 (nucleotide cases)
    assert km.fx_hash(km.mer("TAGCTAG")) == 0xA76409341339D05A
    assert km.fx_hash(km.mer("UGAUGCA", "r")) == 0xDD7C97AE4CA204B4
    assert km.fx_hash(km.mer("", "r")) == 0
    x, y = km.fx_hash(km.mer("TAGCTAG")), km.fx_hash(km.mer("TAGCTAG"), 1)
    assert x != y  # kmer.jl:240-250


def test_transform_known_answers(km, kats, orc):
    # test/runtests.jl:438-485
    g = kats["G10_iscanonical"]
    for t in g["true"]:
        if t:
            assert km.iscanonical(km.mer(t)) is True
    for t in g["false"]:
        assert km.iscanonical(km.mer(t)) is False
    assert str(km.reverse_complement(km.mer("AGCTAGG"))) == "CCTAGCT"
    assert str(km.reverse(km.mer("AGCTAGG"))) == "GGATCGA"
    assert str(km.complement(km.mer("AGCTAGG"))) == "TCGATCC"
    assert str(km.canonical(km.mer("TTGAA"))) == "TTCAA"
    for text, n in kats["G13_gc_count"]["cases"]:                      # test/runtests.jl:1021-1027
        if text:
            assert km.count_gc(km.mer(text.replace("U", "T"))) == n
    assert str(km.to_longsequence(km.mer("TAGCTAGGACA"))) == "TAGCTAGGACA"   # construction.jl:289-324
    rng = np.random.default_rng(1)
    for K in (5, 31, 32, 33, 64):
        ts = [naive.random_text(rng, K) for _ in range(50)]
        arr = km.KmerArray(km.DNAAlphabet[2], K, np.array([naive.kmer_words(t, 2) for t in ts], dtype=np.uint64))
        assert texts(km.reverse_complement(arr)) == [naive.revcomp_text(t) for t in ts]
        # batch as_integer / from_integer (kmer.jl:305-326, :361-384) against the oracle's scalar form
        ints = km.as_integer(arr)
        for row, t in zip(ints, ts):
            val, _ = orc.as_integer(naive.kmer_words(t, 2), K, 2)
            got = int(row) if arr.N == 1 else (int(row[1]) << 64) | int(row[0])
            assert got == val == km.as_integer(km.mer(t))
        dirty = ints.copy()
        if K in (5, 31):    # one word with spare top bits: non-coding bits of u are ignored
            dirty |= np.uint64(0xC000000000000000)
        elif K == 33:       # u128: column 1 is the high half
            dirty[:, 1] |= np.uint64(0xF000000000000000)
        assert km.from_integer(km.DNAAlphabet[2], K, dirty) == arr
        assert km.from_integer(km.DNAAlphabet[2], K, ints) == arr
    with pytest.raises(km.KmersError):
        km.as_integer(km.KmerArray(km.DNAAlphabet[2], 65, np.zeros((1, 3), np.uint64)))


def test_big_chunked_iteration_matches_collect(km, orc):
    """Chunk-buffered iterate() == collect() on a sequence longer than one chunk."""
    L, K = 300_000, 31
    words = orc.synth_words(5, 0, (L * 4 + 63) // 64, 4)
    seq = km.LongSequence(km.DNAAlphabet[4], words, L)
    it = km.CanonicalDNAMers[K](seq)
    it.CHUNK = 1 << 16
    whole = km.collect(it)
    ek, _, _ = orc.canonical(words, L, 4, 2, K)
    assert np.array_equal(whole.words, ek)
    assert sum(1 for _ in it) == L - K + 1
    first = [k.data for _, k in zip(range(70000), it)]
    assert first == [tuple(int(x) for x in r) for r in ek[:70000]]
    km_h, hs = it.collect_with_hashes(seed=3)
    _, eh, _ = orc.canonical(words, L, 4, 2, K, seed=3)
    assert np.array_equal(hs, eh) and km_h == whole


def test_pipelined_iteration_of_all_five_iterators(km, orc):
    """`for x in it` for FwKmers, FwRvIterator, CanonicalKmers, SpacedKmers and UnambiguousKmers over several chunks: chunk c + 1 is
    launched and its copy enqueued before the loop gets chunk c (host._ChunkPipe, the pipeline of julia/KmersHIP.jl's GPUIterator;
    src/iterators/FwKmers.jl:57-66, CanonicalKmers.jl:54-66, SpacedKmers.jl:121-139, UnambiguousKmers.jl:59-62).  Every element
    against the oracle, chunk sizes that divide the iteration and that do not, a loop left early, an EncodeError in the third
    chunk (the elements before it are yielded, then it is thrown), UnambiguousKmers' global starts across chunks."""
    L = 150_003
    words = orc.synth_words(9, 0, (L * 4 + 63) // 64, 4)
    seq = km.LongSequence(km.DNAAlphabet[4], words, L)
    rows = lambda a: [tuple(int(x) for x in r) for r in a]
    for chunk in (1 << 14, 50_000, 1 << 20):
        it = km.FwDNAMers[33](seq)
        it.CHUNK = chunk
        assert [k.data for k in it] == rows(orc.fw_kmers(words, L, 4, 2, 33)[0])
        it = km.FwRvDNAIterator[21](seq)
        it.CHUNK = chunk
        f, r, _ = orc.fwrv(words, L, 4, 2, 21)
        assert [(a.data, b.data) for a, b in it] == list(zip(rows(f), rows(r)))
        it = km.CanonicalDNAMers[31](seq)
        it.CHUNK = chunk
        assert [k.data for k in it] == rows(orc.canonical(words, L, 4, 2, 31)[0])
        it = km.SpacedDNAMers[21, 3](seq)
        it.CHUNK = chunk
        assert [k.data for k in it] == rows(orc.spaced(words, L, 4, 2, 21, 3)[0])
    # a loop left early frees its buffers (the generator's finally) and the next loop starts over
    it = km.CanonicalDNAMers[31](seq)
    it.CHUNK = 1 << 14
    before = it.ctx.pool_stats()["blocks_out"]
    for i, _ in enumerate(it):
        if i == 40_000:
            break
    del _
    import gc
    gc.collect()
    assert it.ctx.pool_stats()["blocks_out"] == before
    # UnambiguousKmers: ambiguity codes sprinkled in, chunks of candidate windows, starts of the WHOLE sequence
    amb = orc.synth_words(11, 0, (L * 4 + 63) // 64, 4, ambig_per_65536=1500)
    aseq = km.LongSequence(km.DNAAlphabet[4], amb, L)
    ek, es, _ = orc.unambiguous(amb, L, 4, 25)
    for chunk in (1 << 14, 61_000):
        it = km.UnambiguousDNAMers[25](aseq)
        it.CHUNK = chunk
        got = list(it)
        assert [k.data for k, _ in got] == rows(ek) and [int(i) for _, i in got] == [int(x) for x in es]
    two = orc.synth_words(12, 0, (L * 2 + 63) // 64, 2)
    it = km.UnambiguousDNAMers[25](km.LongSequence(km.DNAAlphabet[2], two, L))     # a 2-bit source: nothing dropped, HasLength
    it.CHUNK = 1 << 15
    got = list(it)
    assert len(got) == L - 24 and [int(i) for _, i in got] == list(range(1, L - 23))
    assert [k.data for k, _ in got[:1000]] == rows(orc.fw_kmers(two, L, 2, 2, 25)[0][:1000])
    # an EncodeError in the third chunk: everything before it is yielded, then it is thrown at its position
    bad = 2 * (1 << 14) + 5000                      # 0-based symbol index
    poisoned = words.copy()
    w, b = bad * 4 // 64, bad * 4 % 64
    poisoned[w] = (int(poisoned[w]) & ~(0xF << b)) | (0xF << b)     # N
    it = km.CanonicalDNAMers[31](km.LongSequence(km.DNAAlphabet[4], poisoned, L))
    it.CHUNK = 1 << 14
    got = []
    with pytest.raises(km.EncodeError) as e:
        for k in it:
            got.append(k.data)
    assert e.value.position == bad + 1
    assert len(got) == bad - 31 + 1 and got == rows(orc.canonical(words, L, 4, 2, 31)[0][:len(got)])


def test_fused_consumers(km, orc):
    """sketch(fx_hash, CanonicalDNAMers{16}(seq), 1000) (docs/src/minhash.md:34) and the composition
    recipe (docs/src/composition.md:28-39) through the mirror."""
    L = 400_000
    words = orc.synth_words(9, 0, (L * 4 + 63) // 64, 4)
    seq = km.LongSequence(km.DNAAlphabet[4], words, L)
    sk = km.sketch(km.fx_hash, km.CanonicalDNAMers[16](seq), 1000)
    _, eh, _ = orc.canonical(words, L, 4, 2, 16)
    assert np.array_equal(sk, np.unique(eh)[:1000])
    assert np.array_equal(km.sketch(km.fx_hash, km.CanonicalDNAMers[16]("ACGTTGCAAGGCTTACGATCGA"), 1000),
                          np.unique(orc.canonical(naive.ascii_words("ACGTTGCAAGGCTTACGATCGA"), 22, 8, 2, 16)[1]))
    mins = km.minimizers(km.FwDNAMers[8](seq), 20, stride=20)            # test/benchmark.jl:112-119 shape
    exp, _ = orc.minimizers(words, L, 4, 2, 8, 20, 20, 0)
    assert np.array_equal(mins.words, exp)
    comp = km.composition(km.FwDNAMers[4](seq))
    fw, _ = orc.fw_kmers(words, L, 4, 2, 4)
    assert np.array_equal(comp, np.bincount(fw[:, 0].astype(np.int64), minlength=256).astype(np.uint32))


def test_plain_c_client(km, orc, tmp_path):
    """examples/canonical_hashes.c: the C ABI from plain C, checked against the oracle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "kmers.jl_amd", "csrc")
    exe = tmp_path / "canonical_hashes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "canonical_hashes.c"),
                    "-L", csrc, "-lkmers_hip", f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
    text = "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCACGATCTTAGCACTTGCTAGGGATTCGAGGATC"
    out = subprocess.run([str(exe), text], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    ek, eh, _ = orc.canonical(naive.ascii_words(text), len(text), 8, 2, 31)
    assert f"{len(ek)} canonical 31-mers" in out.stdout
    assert f"kmer[0] = 0x{int(ek[0, 0]):016x}  fx_hash = 0x{int(eh[0]):016x}" in out.stdout
    sk = " ".join(f"{int(h):016x}" for h in np.unique(eh)[:8])
    assert sk in out.stdout
    bad = subprocess.run([str(exe), "ACGT" * 10 + "P" + "ACGT" * 10], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "cannot encode 0x50 (Char 'P') at position 41" in bad.stdout


def test_plain_c_batch_client(km, orc, tmp_path):
    """examples/batch_reads.c: kmers_batch / kmers_minhash_batch with KMERS_BATCH_SKIP from plain C, against the oracle."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "kmers.jl_amd", "csrc")
    exe = tmp_path / "batch_reads"
    src = os.path.join(root, "examples", "batch_reads.c")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), src, "-L", csrc, "-lkmers_hip",
                    f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    reads = re.findall(r'^\s+"([ACGTN]+)",', open(src).read(), flags=re.M)
    lines = out.stdout.strip().splitlines()
    assert len(reads) == 4 and len(lines) == 4
    for i, (read, line) in enumerate(zip(reads, lines)):
        n = max(len(read) - 21 + 1, 0)
        hashes, skipped = [], 0
        for j in range(n):                                  # UnambiguousKmers composition, docs/src/faq.md:28-33
            w = read[j:j + 21]
            if "N" in w:
                skipped += 1
                continue
            hashes.append(int(orc.canonical(naive.ascii_words(w), 21, 8, 2, 21)[1][0]))
        sk = sorted(set(hashes))[:8]
        exp = f"read {i}: {n} canonical 21-mers ({skipped} over an ambiguous call), sketch of {len(sk)}:" + \
            "".join(f" {h:016x}" for h in sk)
        assert line == exp


def test_reducer_of_the_reference_benchmark(km, orc):
    """reducer(it) (test/benchmark.jl:9-15) over every iterator type == XOR of the collected elements' head words."""
    rng = np.random.default_rng(12)
    text = naive.random_text(rng, 30_000, p_amb=0.01)
    clean = naive.random_text(rng, 30_000)
    s4, s2 = km.LongDNA[4](text), km.LongDNA[2](clean)

    def fold(arr):
        return int(np.bitwise_xor.reduce(arr.words[:, 0])) if len(arr) else 0
    assert km.reducer(km.FwDNAMers[7](s2)) == fold(km.collect(km.FwDNAMers[7](s2)))
    assert km.reducer(km.FwRvIterator[km.DNAAlphabet[2], 7](s2)) == fold(km.collect(km.FwDNAMers[7](s2)))
    assert km.reducer(km.CanonicalDNAMers[7](s2)) == fold(km.collect(km.CanonicalDNAMers[7](s2)))
    assert km.reducer(km.SpacedDNAMers[7, 5](s2)) == fold(km.collect(km.SpacedDNAMers[7, 5](s2)))
    un = 0
    for kmer, _start in km.collect(km.UnambiguousDNAMers[7](s4)):
        un ^= kmer.data[0]
    assert km.reducer(km.UnambiguousDNAMers[7](s4)) == un
    assert km.reducer(km.CanonicalDNAMers[7](clean)) == fold(km.collect(km.CanonicalDNAMers[7](clean)))   # String source
    with pytest.raises(km.EncodeError):
        km.reducer(km.SpacedDNAMers[7, 5](s4))


def test_wide_kmers_through_the_mirror(km, orc):
    """Kmers of more than four words (K = 150: five words) through the host mirror's consumers and element-wise functions:
    the same calls as with short kmers (Kmer{A,K,N} has no bound on N, src/kmer.jl:97-111)."""
    rng = np.random.default_rng(15)
    K, L = 150, 20_000
    text = naive.random_text(rng, L)
    words = naive.longseq_words(text, 4)
    seq = km.LongDNA[4](text)
    ek, eh, _ = orc.canonical(words, L, 4, 2, K)
    efw, _ = orc.fw_kmers(words, L, 4, 2, K)
    assert km.reducer(km.CanonicalDNAMers[K](seq)) == int(np.bitwise_xor.reduce(ek[:, 0]))
    assert km.reducer(km.FwDNAMers[K](seq)) == int(np.bitwise_xor.reduce(efw[:, 0]))
    assert km.reducer(km.SpacedDNAMers[K, 40](seq)) == int(np.bitwise_xor.reduce(efw[::40, 0]))
    assert np.array_equal(km.sketch(km.fx_hash, km.CanonicalDNAMers[K](seq), 200), np.unique(eh)[:200])
    mins = km.minimizers(km.FwDNAMers[K](seq), 9, stride=50, mode=1)
    exp, _ = orc.minimizers(words, L, 4, 2, K, 9, 50, 1)
    assert np.array_equal(mins.words, exp)
    arr = km.collect(km.FwDNAMers[K](seq))
    assert np.array_equal(arr.words, efw)
    rc = km.reverse_complement(arr)
    assert [str(x) for x in rc[:3]] == [naive.revcomp_text(text[i:i + K]) for i in range(3)]
    assert np.array_equal(km.canonical(arr).words, ek)
    assert np.array_equal(km.fx_hash(km.canonical(arr)), eh)
    recs = [text[:100], text[100:700], text[700:5000]]
    sk = km.sketch_batch(km.fx_hash, km.CanonicalDNAMers[K], recs, 50)
    for r, got in zip(recs, sk):
        want = np.unique(orc.canonical(naive.ascii_words(r), len(r), 8, 2, K)[1])[:50] if len(r) >= K else np.zeros(0, np.uint64)
        assert np.array_equal(got, want)


def test_dispatch_gate_of_the_binding(km, kats):
    """VERDICT r4 (missing 4): the Julia binding's Base.collect goes to the device only from MIN_BASES symbols on
    (julia/KmersHIP.jl, DISPATCH POLICY); the docstring case of src/iterators/FwKmers.jl:14-22 stays the reference's own CPU call.
    The mirror pins the decision function; its explicit gpu_collect runs on the device whatever the length."""
    from kmers_jl_amd import host
    c = kats["G5_fw"]["cases"][0]
    tiny = km.FwDNAMers[3](km.LongDNA[4](c["seq"]))
    assert host.MIN_BASES == 100_000 and not host.gpu_dispatch(tiny)
    assert texts(host.gpu_collect(tiny)) == c["kmers"]          # explicit: on the device even for eight symbols
    rng = np.random.default_rng(3)
    text = "".join(rng.choice(list("ACGT"), host.MIN_BASES))
    assert host.gpu_dispatch(km.CanonicalDNAMers[31](km.LongDNA[4](text)))
    assert not host.gpu_dispatch(km.CanonicalDNAMers[31](km.LongDNA[4](text[:-1])))
    assert host.gpu_dispatch(km.SpacedDNAMers[21, 3](text)) and host.gpu_dispatch(km.UnambiguousDNAMers[31](km.LongDNA[4](text)))
