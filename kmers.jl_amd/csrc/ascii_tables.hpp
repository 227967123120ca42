// ascii_tables.hpp -- byte -> symbol tables for AsciiEncode sources (String / Vector{UInt8}).
//
// encode table = BioSequences.ascii_encode(A, byte) of the KMER's alphabet A (call sites
// src/iterators/FwKmers.jl:123, CanonicalKmers.jl:156, construction_utils.jl:81,229): the symbol's
// encoding, or 0x80 for a byte that is not a symbol of A.  2-bit: ACGT (DNA) / ACGU (RNA), either
// case -> 0..3.  4-bit: the IUPAC letters "-ACMGRSVTWYHKDBN" (U instead of T for RNA), either case
// -> the 4-bit one-hot/OR encoding (index in that string).
// skipping table = the reference's own ASCII_SKIPPING_LUT (src/iterators/common.jl:22-32).
#pragma once
#include <stdint.h>

namespace kmers {

inline void build_ascii_encode_table(int dst_bits, bool rna, uint8_t *t) {
    static const char iupac[] = "-ACMGRSVTWYHKDBN";
    for (int b = 0; b < 256; ++b) t[b] = 0x80;
    for (int v = 0; v < 16; ++v) {
        char c = iupac[v];
        if (c == 'T' && rna) c = 'U';
        uint8_t enc;
        if (dst_bits == 4) {
            enc = (uint8_t)v;
        } else {
            if (v != 1 && v != 2 && v != 4 && v != 8) continue;  // only A, C, G, T/U exist in 2 bits
            enc = (uint8_t)(v == 1 ? 0 : v == 2 ? 1 : v == 4 ? 2 : 3);
        }
        t[(uint8_t)c] = enc;
        if (c != '-') t[(uint8_t)(c + 32)] = enc;  // lower case
    }
}

inline void build_ascii_skipping_table(uint8_t *t) {
    for (int b = 0; b < 256; ++b) t[b] = 0xff;
    const char *acgt[4] = {"Aa", "cC", "gG", "TtUu"};
    for (int i = 0; i < 4; ++i)
        for (const char *p = acgt[i]; *p; ++p) t[(uint8_t)*p] = (uint8_t)i;
    for (const char *p = "-MRSVWYHKDBN"; *p; ++p) {
        t[(uint8_t)*p] = 0xf0;
        if (*p != '-') t[(uint8_t)(*p + 32)] = 0xf0;
    }
}

}  // namespace kmers
