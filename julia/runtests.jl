#!/usr/bin/env julia
# runtests.jl -- the parity tests a Kmers.jl maintainer runs ON A HOST WITH JULIA AND AN MI355X: every route of julia/KmersHIP.jl
# against Kmers.jl's own iteration of the same iterator, element for element (`===` on isbits values), in the style of the
# reference's test/runtests.jl (`:78-100`, `:154-169`, `:674-690`, `:739-761`, `:774-787`, `:857-869`, `:903-910`, `:916-945`).
#
# NOT EXECUTED IN THIS REPOSITORY'S CI: `julia` is not part of the build image (DESIGN.md).  What runs here instead is the same
# mapping through ctypes (kmers.jl_amd/host.py under tests/test_gpu_*.py, against oracle/), and tests/test_julia_binding.py, which
# parses julia/KmersHIP.jl against include/kmers_hip.h and checks that every KmersHIP name this file uses exists there.
#
#     julia --project=/path/to/Kmers.jl julia/runtests.jl          (KMERS_HIP_LIB=/path/to/libkmers_hip.so if it is not in-tree)
using Test, Random
using Kmers, BioSequences
include(joinpath(@__DIR__, "KmersHIP.jl"))
using .KmersHIP

# Kmers.jl's own method, whatever KmersHIP.MIN_BASES[] says (the overload of Base.collect is gated: KmersHIP.jl, DISPATCH POLICY)
cpu(it) = invoke(collect, Tuple{Any}, it)
# the three device routes of an iterator: the bulk call, Base.collect above the gate, the chunk-buffered iterate()
function routes(it; chunk = 4099)
    old = KmersHIP.MIN_BASES[]
    KmersHIP.MIN_BASES[] = 0
    try
        return (KmersHIP.gpu_collect(it), collect(it), collect(KmersHIP.gpu(it; chunk = chunk)))
    finally
        KmersHIP.MIN_BASES[] = old
    end
end
function same(it; kw...)
    want = cpu(it)
    for got in routes(it; kw...)
        eltype(got) == eltype(want) || return false
        length(got) == length(want) || return false
        all(got[i] === want[i] for i in eachindex(want)) || return false
    end
    return true
end

const RNG = Random.Xoshiro(20261004)
rand_dna4(n) = randdnaseq(RNG, n)                                   # LongDNA{4}, no ambiguous symbols
rand_dna2(n) = LongDNA{2}(randdnaseq(RNG, n))
rand_rna4(n) = randrnaseq(RNG, n)
function with_ns(s::LongDNA{4}, share)                              # an N at a share of the positions
    t = copy(s)
    for i in eachindex(t)
        rand(RNG) < share && (t[i] = DNA_N)
    end
    return t
end

@testset "KmersHIP against Kmers.jl" begin

@testset "FwKmers: sources x kmer alphabets x K (FwKmers.jl:57-115; runtests.jl:674-690)" begin
    for n in (0, 1, 30, 31, 1000, 70_001), K in (1, 3, 21, 31, 32, 33, 63, 64, 65, 127)
        s4, s2, r4 = rand_dna4(n), rand_dna2(n), rand_rna4(n)
        @test same(FwDNAMers{K}(s4))                                # FourToTwo
        @test same(FwDNAMers{K}(s2))                                # Copyable
        @test same(FwRNAMers{K}(r4))
        @test same(FwKmers{DNAAlphabet{4}, K}(s4))                  # Copyable, 4-bit kmers
        @test same(FwKmers{DNAAlphabet{4}, K}(s2))                  # TwoToFour
        @test same(FwDNAMers{K}(String(s4)))                        # AsciiEncode
        @test same(FwDNAMers{K}(codeunits(String(s4))))
        @test same(FwRNAMers{K}(String(r4)))
        n >= 2 && @test same(FwDNAMers{K}(view(s4, 2:n)))            # LongSubSeq: kmers_seq.first_base (runtests.jl:162)
        n >= 19 && @test same(FwDNAMers{K}(view(s2, 18:n)))
    end
    @test same(FwDNAMers{21}(collect(rand_dna4(5000))))              # Vector{DNA}: GenericRecoding (construction.jl:90-98)
    @test same(FwKmers{DNAAlphabet{4}, 9}(with_ns(rand_dna4(5000), 0.05)))   # 4-bit kmers keep ambiguous symbols (runtests.jl:857)
end

@testset "FwRvIterator and CanonicalKmers (CanonicalKmers.jl:54-225; runtests.jl:739-761)" begin
    for n in (0, 31, 1000, 70_001), K in (1, 5, 16, 31, 32, 33, 63, 64, 65)
        s4, s2 = rand_dna4(n), rand_dna2(n)
        for s in (s4, s2, String(s4))
            @test same(FwRvIterator{DNAAlphabet{2}, K}(s))
            @test same(CanonicalDNAMers{K}(s))
        end
        @test same(CanonicalKmers{DNAAlphabet{4}, K}(s4))
        @test same(FwRvIterator{DNAAlphabet{4}, K}(s2))
        @test same(CanonicalKmers{RNAAlphabet{2}, K}(rand_rna4(n)))
        n >= 8 && @test same(CanonicalDNAMers{K}(view(s4, 8:n)))
    end
    s = rand_dna4(200_000)
    v, h = KmersHIP.collect_with_hashes(CanonicalDNAMers{31}(s))
    want = cpu(CanonicalDNAMers{31}(s))
    @test v == want && h == map(fx_hash, want)                      # kmer.jl:255-261
    v, h = KmersHIP.collect_with_hashes(CanonicalDNAMers{63}(s); seed = UInt(0x1234))
    @test h == [fx_hash(x, UInt(0x1234)) for x in cpu(CanonicalDNAMers{63}(s))]
end

@testset "SpacedKmers, strict (SpacedKmers.jl:83-139; runtests.jl:857-869)" begin
    for n in (0, 21, 1000, 70_001), (K, J) in ((21, 3), (3, 3), (5, 7), (31, 1), (33, 40), (8, 64))
        s4, s2 = rand_dna4(n), rand_dna2(n)
        @test same(SpacedDNAMers{K, J}(s4))
        @test same(SpacedDNAMers{K, J}(s2))
        @test same(SpacedKmers{DNAAlphabet{4}, K, J}(s4))
        @test same(SpacedDNAMers{K, J}(String(s4)))
        n >= 5 && @test same(SpacedDNAMers{K, J}(view(s4, 5:n)))
    end
    @test same(each_codon(rand_rna4(30_000)))
    # J >= K never looks at the symbols between the kmers (SpacedKmers.jl:133-134)
    s = rand_dna4(10_000)
    s[6] = DNA_N
    @test same(SpacedDNAMers{5, 7}(s))
end

@testset "the first ambiguous symbol is an EncodeError, after the elements in front of it (FwKmers.jl:112; runtests.jl:868-869)" begin
    for pos in (1, 17, 31, 32, 5000, 69_990)
        s = rand_dna4(70_000)
        s[pos] = DNA_M
        for it in (FwDNAMers{31}(s), CanonicalDNAMers{31}(s), FwRvIterator{DNAAlphabet{2}, 31}(s), SpacedDNAMers{21, 3}(s))
            err = try cpu(it); nothing catch e; e end
            @test err isa BioSequences.EncodeError
            for f in (KmersHIP.gpu_collect, x -> collect(KmersHIP.gpu(x; chunk = 1000)))
                got = try f(it); nothing catch e; e end
                @test got isa BioSequences.EncodeError && sprint(showerror, got) == sprint(showerror, err)
            end
            # iterate() yields what Kmers.jl's loop yields before it throws
            seen_cpu, seen_gpu = eltype(it)[], eltype(it)[]
            try for x in it; push!(seen_cpu, x); end catch; end
            try for x in KmersHIP.gpu(it; chunk = 1000); push!(seen_gpu, x); end catch; end
            @test seen_gpu == seen_cpu
        end
        t = String(s)
        @test_throws BioSequences.EncodeError KmersHIP.gpu_collect(FwDNAMers{31}(t))
    end
end

@testset "UnambiguousKmers: (kmer, start), ambiguous symbols skipped (UnambiguousKmers.jl:59-148; runtests.jl:774-787)" begin
    for n in (0, 30, 1000, 70_001, 300_000), K in (1, 4, 21, 31, 32, 33, 64), share in (0.0, 0.01, 0.04, 0.3)
        s = with_ns(rand_dna4(n), share)
        @test same(UnambiguousDNAMers{K}(s))
        @test same(UnambiguousDNAMers{K}(String(s)))                # ASCII_SKIPPING_LUT (UnambiguousKmers.jl:109-132)
        n >= 3 && @test same(UnambiguousDNAMers{K}(view(s, 3:n)))    # starts are positions of the VIEW
    end
    @test same(UnambiguousDNAMers{21}(rand_dna2(50_000)))            # 2-bit source: HasLength, nothing dropped (:33-37)
    @test same(UnambiguousRNAMers{9}(rand_rna4(50_000)))
    # the skip variant of the C5 configuration: the lattice elements of UnambiguousDNAMers{21} (SURVEY.md 8a)
    s = with_ns(rand_dna4(100_000), 0.04)
    want = [(k, i) for (k, i) in cpu(UnambiguousDNAMers{21}(s)) if (i - 1) % 3 == 0]
    @test [(k, i) for (k, i) in KmersHIP.gpu_collect(UnambiguousDNAMers{21}(s)) if (i - 1) % 3 == 0] == want
end

@testset "Base.collect is gated (FwKmers.jl:14-22: tiny inputs keep Kmers.jl's own method)" begin
    old = KmersHIP.MIN_BASES[]
    KmersHIP.MIN_BASES[] = 100_000
    @test !KmersHIP.gpu_dispatch(FwDNAMers{3}(dna"AGCGTATA")) && collect(FwDNAMers{3}(dna"AGCGTATA")) == cpu(FwDNAMers{3}(dna"AGCGTATA"))
    @test KmersHIP.gpu_dispatch(FwDNAMers{3}(rand_dna4(100_000)))
    KmersHIP.MIN_BASES[] = typemax(Int)
    @test !KmersHIP.gpu_dispatch(FwDNAMers{3}(rand_dna4(200_000)))
    KmersHIP.MIN_BASES[] = old
end

@testset "fused consumers (docs/src/minhash.md:31-41, composition.md:28-39, replacements.md:33-51)" begin
    s = rand_dna4(500_000)
    it = CanonicalDNAMers{16}(s)
    want = sort!(unique!(map(fx_hash, cpu(it))))
    @test KmersHIP.sketch(fx_hash, it, 1000) == want[1:1000]
    short = rand_dna4(300)                                           # fewer than s distinct values: all of them (res.n_out)
    @test KmersHIP.sketch(fx_hash, CanonicalDNAMers{16}(short), 1000) == sort!(unique!(map(fx_hash, cpu(CanonicalDNAMers{16}(short)))))
    for K in (1, 4, 8)
        counts = zeros(UInt32, 4^K)
        for kmer in FwDNAMers{K}(s)
            counts[as_integer(kmer) + 1] += 1
        end
        @test KmersHIP.composition(FwDNAMers{K}(s)) == counts
    end
    # the true sliding-window minimum of fx_hash over W consecutive kmers (mode = 1)
    K, W = 8, 5
    kmers = cpu(FwDNAMers{K}(s))
    want = [kmers[i:(i + W - 1)][argmin(map(fx_hash, kmers[i:(i + W - 1)]))] for i in 1:(length(kmers) - W + 1)]
    @test KmersHIP.minimizers(FwDNAMers{K}(s), W; mode = 1) == want
end

@testset "element-wise functions over vectors of kmers (kmer.jl:255-261, :305-326; transformations.jl:1-41; runtests.jl:903-945)" begin
    @test fx_hash([mer"TAGCTAG"d]) == [0xa76409341339d05a]            # the reference's own known answers (runtests.jl:903-910)
    @test fx_hash([mer"UGAUGCA"r]) == [0xdd7c97ae4ca204b4]
    for K in (1, 7, 31, 32, 33, 63, 64)
        v = cpu(FwDNAMers{K}(rand_dna4(20_000)))
        @test fx_hash(v) == map(fx_hash, v)
        @test fx_hash(v, UInt(99)) == [fx_hash(x, UInt(99)) for x in v]
        @test KmersHIP.reverse_each(v) == map(reverse, v) && reverse(v) == v[end:-1:1]   # (Base.reverse of a Vector keeps its meaning)
        @test BioSequences.complement(v) == map(BioSequences.complement, v)
        @test BioSequences.reverse_complement(v) == map(BioSequences.reverse_complement, v)
        @test BioSequences.canonical(v) == map(BioSequences.canonical, v)
        @test as_integer(v) == map(as_integer, v)
        v4 = cpu(FwKmers{DNAAlphabet{4}, K}(with_ns(rand_dna4(20_000), 0.1)))
        @test BioSequences.reverse_complement(v4) == map(BioSequences.reverse_complement, v4)
    end
end

@testset "batches of records: one call for many sequences (docs/src/minhash.md:31-35, faq.md:28-33)" begin
    seqs = [rand_dna4(rand(RNG, 0:400)) for _ in 1:5000]
    kmers, hs, offsets = KmersHIP.collect_batch(CanonicalDNAMers{31}, seqs; hashes = true)
    @test length(offsets) == length(seqs) + 1 && offsets[1] == 0
    for (i, s) in enumerate(seqs)
        want = cpu(CanonicalDNAMers{31}(s))
        got = kmers[(offsets[i] + 1):offsets[i + 1]]
        @test got == want && hs[(offsets[i] + 1):offsets[i + 1]] == map(fx_hash, want)
    end
    kmers, _, offsets = KmersHIP.collect_batch(FwDNAMers{21}, seqs)
    @test all(kmers[(offsets[i] + 1):offsets[i + 1]] == cpu(FwDNAMers{21}(s)) for (i, s) in enumerate(seqs))
    kmers, offsets = KmersHIP.collect_batch(SpacedDNAMers{21, 3}, seqs)
    @test all(kmers[(offsets[i] + 1):offsets[i + 1]] == cpu(SpacedDNAMers{21, 3}(s)) for (i, s) in enumerate(seqs))
    genomes = [rand_dna4(rand(RNG, 5_000:15_000)) for _ in 1:200]
    sk = KmersHIP.sketch_batch(fx_hash, CanonicalDNAMers{16}, genomes, 1000)
    @test all(sk[i] == sort!(unique!(map(fx_hash, cpu(CanonicalDNAMers{16}(g)))))[1:min(1000, end)] for (i, g) in enumerate(genomes))
end

@testset "iterate(): an early exit frees the pipeline, a second loop starts over (FwKmers.jl:57-66)" begin
    s = rand_dna4(1_000_000)
    g = KmersHIP.gpu(CanonicalDNAMers{31}(s); chunk = 10_000)
    first100 = eltype(g)[]
    for x in g
        push!(first100, x)
        length(first100) == 100 && break
    end
    @test first100 == cpu(CanonicalDNAMers{31}(s))[1:100]
    @test collect(g) == cpu(CanonicalDNAMers{31}(s))
    @test length(g) == length(CanonicalDNAMers{31}(s)) && eltype(g) == eltype(CanonicalDNAMers{31}(s))
    @test Base.IteratorSize(typeof(KmersHIP.gpu(UnambiguousDNAMers{5}(s)))) == Base.SizeUnknown()
end

end
