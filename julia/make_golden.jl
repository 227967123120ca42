#!/usr/bin/env julia
# Regenerates known-answer vectors FROM THE REAL REFERENCE, in the layout of tests/golden/kats.json,
# for maintainers with Julia + Kmers.jl + BioSequences.jl (absent from the build image, where
# kats.json was written by hand from the reference's tests and docstrings).  It also emits the one
# fact kats.json can only pin structurally: the LongSequence word order (SURVEY.md section 8c).
#
#     julia --project=/path/to/Kmers.jl julia/make_golden.jl > /tmp/kats_from_julia.json
#     python tests/golden/compare_with_julia.py /tmp/kats_from_julia.json
using Kmers, BioSequences

hexs(t) = "[" * join(("\"0x$(string(w, base = 16, pad = 16))\"" for w in t), ", ") * "]"
words(k::Kmer) = hexs(k.data)

function main()
    println("{")
    # LongSequence word order: the packed words of a 40-symbol sequence in both alphabets
    text = "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCACGATC"
    println("  \"longseq_words\": {\"text\": \"$text\", \"dna2\": $(hexs(LongDNA{2}(text).data)), ",
            "\"dna4\": $(hexs(LongDNA{4}(text).data))},")
    # kmer memory layout + fx_hash + as_integer for a few K around the word boundaries
    print("  \"kmers\": [")
    first = true
    for K in (1, 7, 21, 31, 32, 33, 40)
        for (name, T) in (("dna2", DNAKmer{K}), ("dna4", Kmer{DNAAlphabet{4}, K}))
            k = T(text[1:K])
            first || print(", ")
            first = false
            print("\n    {\"alphabet\": \"$name\", \"text\": \"$(text[1:K])\", \"words\": $(words(k)), ",
                  "\"fx_hash\": \"0x$(string(Kmers.fx_hash(k), base = 16, pad = 16))\", ",
                  "\"revcomp\": $(words(reverse_complement(k))), \"canonical\": $(words(canonical(k)))}")
        end
    end
    println("\n  ],")
    # iterators over the same text (4-bit source, 2-bit kmers): first three elements as words
    seq = LongDNA{4}(text)
    println("  \"canonical31_first3\": [", join((words(k) for k in Iterators.take(CanonicalDNAMers{31}(seq), 3)), ", "), "],")
    println("  \"fwrv33_first2\": [", join(("[" * words(a) * ", " * words(b) * "]" for (a, b) in Iterators.take(FwRvIterator{DNAAlphabet{2}, 33}(seq), 2)), ", "), "],")
    amb = dna"TAGCWSAGACYWNACGCNACG--"
    println("  \"unambiguous3\": {\"seq\": \"$(amb)\", \"items\": [", join(("[$(words(k)), $i]" for (k, i) in UnambiguousDNAMers{3}(amb)), ", "), "]}")
    println("}")
end

main()
