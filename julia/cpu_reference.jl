#!/usr/bin/env julia
# CPU baseline with the REAL reference: times Kmers.jl's own iterators on the same synthetic
# packed input bench.py uses (SURVEY.md section 8d).  Julia is not in the build image, so this
# script has not been run there; bench.py's `cpu_baseline` is the C restatement under oracle/
# ("kind": "port").  On a host with Julia + Kmers.jl + BioSequences.jl:
#
#     julia --project=/path/to/Kmers.jl julia/cpu_reference.jl [n_bases=100000000] [K=31] [bits=4]
#
# prints one JSON line comparable with bench.py's cpu_baseline object, plus the XOR folds that
# tests/golden/synth_10k.json holds for n_bases = 10000 (so the generator and the real iterators
# can be pinned against this build's fixture in one go).
using Kmers, BioSequences

const GOLDEN = 0x9E3779B97F4A7C15

# SplitMix64 finaliser on a counter: word idx of the stream `seed` (oracle/kmers_oracle.c, orc_synth_rand64)
function rand64(seed::UInt64, idx::UInt64)
    z = seed + (idx + 0x01) * GOLDEN
    z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
    z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
    z ⊻ (z >> 31)
end

# LongSequence.data words of the synthetic sequence: 2 random bits per base, little-endian symbols
function synth_words(seed::UInt64, n_words::Int, bits::Int)
    out = Vector{UInt64}(undef, n_words)
    for k in 0:(n_words - 1)
        w = UInt64(k)
        if bits == 2
            out[k + 1] = rand64(seed, w)
        else
            r = rand64(seed, w >> 1) >> (32 * (w & 1))
            word = UInt64(0)
            for j in 0:15
                word |= (UInt64(1) << ((r >> (2j)) & 3)) << (4j)
            end
            out[k + 1] = word
        end
    end
    out
end

function main()
    n_bases = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 100_000_000
    K = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 31
    bits = length(ARGS) >= 3 ? parse(Int, ARGS[3]) : 4
    seed = GOLDEN ⊻ UInt64(2)
    n_words = cld(n_bases * bits, 64)
    data = synth_words(seed, n_words, bits)
    # the (data, len) constructor the reference itself uses (src/construction.jl:299)
    seq = LongSequence{DNAAlphabet{bits}}(data, UInt(n_bases))
    function pass(seq, ::Val{K}) where {K}
        kx = UInt64(0); hx = UInt64(0); n = 0
        for kmer in CanonicalDNAMers{K}(seq)         # src/iterators/CanonicalKmers.jl:199-225
            kx ⊻= kmer.data[end]                     # low word of the kmer
            hx ⊻= Kmers.fx_hash(kmer)                # src/kmer.jl:255-261
            n += 1
        end
        (kx, hx, n)
    end
    pass(LongSequence{DNAAlphabet{bits}}(data[1:min(end, 64)], UInt(min(n_bases, 64 * 64 ÷ bits))), Val(K))  # compile
    t = @elapsed (kx, hx, n) = pass(seq, Val(K))
    println("{\"value\": $(n_bases / t / 1e9), \"unit\": \"Gbases/s\", \"cores\": 1, \"kind\": \"reference\", ",
            "\"sample\": \"CanonicalDNAMers{$K} + fx_hash over $n_bases bases, $(bits)-bit source\", ",
            "\"n\": $n, \"low_word_xor\": \"0x$(string(kx, base = 16, pad = 16))\", ",
            "\"hash_xor\": \"0x$(string(hx, base = 16, pad = 16))\"}")
end

main()
