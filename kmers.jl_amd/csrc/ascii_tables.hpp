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

// The same tables WITHOUT memory: entry `c` of table `table` computed from two immediates, so that a kernel
// fills its 256-byte LDS copy with no load from HBM in front of its first barrier (the byte loads of the table
// used to add a whole global-load latency to every short-lived workgroup of the byte-source kernels).
//   table 0 / 1: ascii_encode of the 2-bit DNA / RNA alphabet   2 / 3: of the 4-bit DNA / RNA alphabet
//   table 4    : ASCII_SKIPPING_LUT
//   table 5 / 6: NOT text -- one BioSymbols value per byte (Vector{DNA} / Vector{RNA} and other collections of
//                nucleotide symbols: the reference's GenericRecoding, src/construction.jl:90-98): BioSequences.encode of
//                the 2-bit alphabets (one-hot values only -> trailing_zeros) / of the 4-bit alphabets (every value < 16)
#ifdef __HIPCC__
#define KMERS_HD __host__ __device__
#else
#define KMERS_HD
#endif
constexpr int ASCII_TABLE_SKIPPING = 4;
constexpr int SYMBOL_TABLE_2BIT = 5, SYMBOL_TABLE_4BIT = 6;
// UnambiguousKmers over a collection of symbols (the generic method, UnambiguousKmers.jl:88-106): an ambiguous symbol
// (count_ones > 1) is skipped (0xf0), a certain one shifted in, and the gap fails in `shift` -> encode (0xff: EncodeError)
constexpr int SYMBOL_TABLE_SKIPPING = 7;
KMERS_HD inline uint8_t ascii_entry(uint32_t table, uint32_t c) {
    if (table >= (uint32_t)SYMBOL_TABLE_2BIT) {  // symbol values, not letters
        if (table == (uint32_t)SYMBOL_TABLE_SKIPPING) {
            if (c == 0u || c > 15u) return 0xff;
            if (c & (c - 1u)) return 0xf0;
            return (uint8_t)(c == 1u ? 0u : c == 2u ? 1u : c == 4u ? 2u : 3u);
        }
        if (c > 15u) return 0x80;
        if (table == (uint32_t)SYMBOL_TABLE_4BIT) return (uint8_t)c;
        if (c == 0 || (c & (c - 1u))) return 0x80;  // gap or ambiguous: EncodeError (count_ones != 1)
        return (uint8_t)(c == 1u ? 0u : c == 2u ? 1u : c == 4u ? 2u : 3u);
    }
    // IUPAC value ("-ACMGRSVTWYHKDBN" index) of the letters a..p and q..z, one nibble each; t and u hold 8
    //                  p o n m l k j i h g f e d c b a                      z y x w v u t s r q
    const uint64_t LO = 0x00F30C00B400D2E1ull, HI = 0x0000000A09788650ull >> 0;
    const uint32_t lower = c | 0x20u;
    const bool letter = lower >= 0x61u && lower <= 0x7Au && (c & 0x40u);
    uint32_t v = 0;
    if (letter) {
        const uint32_t idx = lower - 0x61u;
        v = idx < 16u ? (uint32_t)(LO >> (4u * idx)) & 15u : (uint32_t)(HI >> (4u * (idx - 16u))) & 15u;
        const bool rna = (table & 1u) != 0;
        if (table != (uint32_t)ASCII_TABLE_SKIPPING) {
            if (lower == 0x74u && rna) v = 0;   // 't' is not an RNA symbol
            if (lower == 0x75u && !rna) v = 0;  // 'u' is not a DNA symbol
        }
    }
    const bool one_hot = v != 0 && (v & (v - 1u)) == 0;
    const uint32_t code = v == 1u ? 0u : v == 2u ? 1u : v == 4u ? 2u : 3u;  // trailing_zeros of a one-hot nibble
    if (table == (uint32_t)ASCII_TABLE_SKIPPING) {
        if (c == 0x2Du) return 0xf0;            // '-'
        if (v == 0) return 0xff;
        return one_hot ? (uint8_t)code : (uint8_t)0xf0;
    }
    if (table >= 2u) {                           // 4-bit alphabets: every IUPAC letter and the gap
        if (c == 0x2Du) return 0;
        return v ? (uint8_t)v : (uint8_t)0x80;
    }
    return one_hot ? (uint8_t)code : (uint8_t)0x80;
}

}  // namespace kmers
