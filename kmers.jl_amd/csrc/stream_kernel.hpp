// stream_kernel.hpp -- the tile kernel behind kmers_fw / kmers_canonical / kmers_spaced /
// kmers_reduce_xor (reference: src/iterators/FwKmers.jl:57-115, CanonicalKmers.jl:54-144,
// :220-225, SpacedKmers.jl:83-139, src/kmer.jl:255-261).
//
// One workgroup (256 threads = 4 wavefronts of 64) owns a tile of `tile_kmers` consecutive
// kmer indices.  Phase 1 reads the source words the tile touches once (coalesced 8-byte
// loads), turns 4-bit one-hot symbols into 2-bit codes (validating count_ones == 1) and
// stages the 2-bit little-endian stream -- the tile plus its (K-1)-base overlap into the
// next tile -- in LDS.  Phase 2 cuts every window out of that stream with a 64-bit funnel
// shift, forms forward / reverse-complement / canonical / fx_hash in registers and writes
// 16 bytes per lane per store instruction, lanes consecutive, so every wave store covers
// whole 128-byte lines.  HBM-bound on the output: see DESIGN.md for bytes per kmer.
#pragma once
#include "ascii_tables.hpp"
#include "device_bits.hpp"

namespace kmers {

#ifndef KMERS_BLOCK
#define KMERS_BLOCK 256
#endif
constexpr int BLOCK = KMERS_BLOCK;             // threads per workgroup (multiple of 64)
constexpr int WAVES = BLOCK / 64;
#ifndef KMERS_MAX_TILE_BITS
#define KMERS_MAX_TILE_BITS 32768
#endif
constexpr int MAX_TILE_BITS = KMERS_MAX_TILE_BITS;           // LDS stream bits a tile may span (excl. overlap): 16384 2-bit symbols
constexpr int MAX_TILE_BASES = MAX_TILE_BITS / 2;
constexpr int LDS_QWORDS = MAX_TILE_BITS / 64 + 16;

enum Mode { MODE_FW = 0, MODE_CANON = 1, MODE_XOR = 2, MODE_SKETCH = 3, MODE_COUNT = 4, MODE_MINIMIZER = 5 };

struct StreamArgs {
    const uint64_t *src;     // LongSequence.data in HBM
    uint64_t first_bit;      // bit offset of symbol 1 inside src
    uint64_t n_bases;        // symbols in the view
    uint64_t n_kmers;        // elements to produce
    uint64_t inspect_end;    // symbols [0, inspect_end) are validated (see kernel)
    uint64_t *out_a;         // FW: forward kmers   CANON: canonical kmers (nullable)  XOR: accumulator
    uint64_t *out_b;         // FW: reverse complements (nullable)   CANON: hashes (nullable)
    uint64_t seed;           // fx_hash seed
    unsigned long long *err_slot;  // atomicMin of error_key(first offending symbol): position (0-based + err_origin) << 8 | raw symbol
    uint64_t err_origin;           // kmers_seq.index_origin (+ the chunk offset of chunked launches)
    uint64_t n_tiles;
    uint32_t k;
    uint32_t stride;
    uint32_t tile_kmers;
    uint32_t subtiles;       // strided kernels: consecutive tiles per workgroup, the next one's source words in flight (0 = 1)
    uint32_t split_order;    // 1: workgroups visit the two halves of the tile range alternately (see the kernel)
    uint32_t xor_canonical;  // MODE_XOR: 1 = canonical kmers, 0 = forward kmers
    uint64_t *stamps;        // diagnostic builds (-DKMERS_STAMPS) only: per-workgroup s_memrealtime stamps
    uint32_t ascii_table;     // SRC_BITS == 8: which byte -> symbol table (ascii_entry(), ascii_tables.hpp)
    // MODE_SKETCH: hashes below the threshold (threshold_ptr[0] when non-NULL, else `threshold`) are appended
    //              to out_a[0..capacity) through the counter out_b[0]
    // MODE_COUNT : out_a = uint32 counts[4^K] indexed by as_integer(forward kmer), global atomics
    uint64_t threshold;
    uint64_t capacity;
    const uint64_t *threshold_ptr;  // MODE_SKETCH, device-resident sketch: the running threshold in HBM
    const uint64_t *best;           // run kernel, device-resident sketch: the current bottom-s set (ascending) ...
    const uint64_t *best_n_ptr;     // ... and its size; hashes already in it are not candidates again
    uint64_t *recent;               // MODE_SKETCH: direct-mapped table of recently appended hashes (see sketch_candidate)
    uint32_t recent_mask;           // its size - 1 (a power of two)
    // MODE_MINIMIZER: window = `window_kmers` consecutive kmers per element, elements `stride` apart;
    //                 minimizer_mode 0 = the reference's published example, 1 = true sliding-window minimum
    uint32_t window_kmers;
    uint32_t minimizer_mode;
    uint32_t tuples;          // 1: array-of-structs output in out_a (Tuple{Kmer,Kmer} / Tuple{Kmer,UInt64}), out_b unused
    long long *out_starts;    // MODE_FW: 1-based start of every kmer (+ start_origin): UnambiguousKmers over a clean sequence
    uint64_t start_origin;
};

// First inspected offending symbol of one source word -> err_slot (rare path, kept inline and
// call-free so the kernel needs no stack).  `f` holds one flag per symbol at bit SRC_BITS*j.
// Inspected set = what the reference's iterate() would have looked at before stopping: every
// symbol below inspect_end, except (stride >= K) the gaps between kmers
// (src/iterators/SpacedKmers.jl:133-134).
// The slot receives error_key = (global 0-based position << 8) | raw source encoding, so that the host can build the
// reference's EncodeError (construction.jl:108-110) from the slot alone: nothing is re-read from the sequence after the
// launch (the caller may have freed it; several asynchronous launches may share one slot -- the smallest position wins).
__device__ __forceinline__ unsigned long long error_key(uint64_t pos, uint64_t enc) { return (pos << 8) | (enc & 0xffull); }

template <int SRC_BITS, bool STRIDE1>
__device__ __forceinline__ void report_bad_symbols(unsigned long long *err_slot, uint64_t first_bit,
                                                   uint64_t inspect_end, uint32_t stride, uint32_t k,
                                                   uint64_t word_index, uint64_t f, uint64_t x, uint64_t origin) {
    constexpr int PER = 64 / SRC_BITS;
    // symbol index of symbol 0 of this word; negative inside the first word of an offset view
    // (both terms are multiples of SRC_BITS, so the division is exact)
    const long long base0 = ((long long)(word_index * 64) - (long long)first_bit) / SRC_BITS;
    if (base0 < 0) f &= ~0ull << (uint32_t)(-base0 * SRC_BITS);
    const long long room = (long long)inspect_end - base0;  // symbols of this word below inspect_end
    if (room <= 0) return;
    if (room < PER) f &= (1ull << (uint32_t)(room * SRC_BITS)) - 1ull;
    if constexpr (!STRIDE1) {  // stride 1 has no gaps
        if (stride >= k) {
#pragma unroll 1
            for (uint32_t j = 0; j < (uint32_t)PER; ++j)
                if (((f >> (SRC_BITS * j)) & 1ull) && ((uint64_t)(base0 + j) % stride) >= k) f &= ~(1ull << (SRC_BITS * j));
        }
    }
    if (f) {
        const uint32_t j = (uint32_t)(__ffsll((long long)f) - 1) / SRC_BITS;  // first offending symbol of the word
        atomicMin(err_slot, error_key((uint64_t)(base0 + (long long)j) + origin, (x >> (SRC_BITS * j)) & ((1ull << SRC_BITS) - 1ull)));
    }
}

// 4-bit source: flags of the symbols with count_ones != 1, from pack_4to2's `bad`
__device__ __forceinline__ uint64_t flags_from_bad4(uint64_t bad) {
    return (bad | (bad >> 1) | (bad >> 2)) & 0x1111111111111111ull;
}

// ---- symbol-level helpers generic in the kmer alphabet width (DST = 2 or 4 bits) -------------
// reverse the order of the DST-bit symbols of one word (BioSequences.reversebits)
template <int DST>
__device__ __forceinline__ uint64_t rev_symbols(uint64_t x) {
    uint64_t r = rev2(x);
    if constexpr (DST == 4) r = ((r >> 2) & 0x3333333333333333ull) | ((r & 0x3333333333333333ull) << 2);
    return r;
}
// complement of every symbol of a word: 2-bit NOT; 4-bit = bit reversal inside each nibble
// (A=0001<->T=1000, C=0010<->G=0100; gap and N are fixed points) -- complement_bitpar
template <int DST>
__device__ __forceinline__ uint64_t comp_symbols(uint64_t x) {
    if constexpr (DST == 2) return ~x;
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
    return ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
}
// 2-bit codes of 16 symbols -> 16 one-hot nibbles (TwoToFour: 1 << code, construction_utils.jl:35)
__device__ __forceinline__ uint64_t expand_2to4(uint32_t x) {
    uint64_t t = x;
    t = (t | (t << 16)) & 0x0000FFFF0000FFFFull;
    t = (t | (t << 8)) & 0x00FF00FF00FF00FFull;
    t = (t | (t << 4)) & 0x0F0F0F0F0F0F0F0Full;
    t = (t | (t << 2)) & 0x3333333333333333ull;  // code c in the low two bits of every nibble
    const uint64_t M1 = 0x1111111111111111ull;
    uint64_t c0 = t & M1, c1 = (t >> 1) & M1, n0 = c0 ^ M1, n1 = c1 ^ M1;
    return (n1 & n0) | ((n1 & c0) << 1) | ((c1 & n0) << 2) | ((c1 & c0) << 3);
}

// forward / reverse-complement kmers (N words, head word first) of the window whose first
// symbol sits at bit `bit` of the LDS stream (DST bits per symbol, little-endian by symbol).
//   W  = K*DST stream bits, symbol j of the window at bits DST*j
//   rc = comp(W)                     (the complement of symbol j belongs at big-endian slot K-1-j)
//   fw = symbol-reversal(W) >> (64N - K*DST)
// which equals K applications of shift_encoding / shift_first_encoding
// (construction_utils.jl:129-134, kmer.jl:511-518) to a zero kmer.
template <int N, int DST>
__device__ __forceinline__ void window(const uint64_t *lds, uint32_t bit, uint32_t k, uint64_t mask,
                                       uint64_t (&fw)[N], uint64_t (&rc)[N]) {
    const uint32_t q = bit >> 6, s = bit & 63u;
    uint64_t W[N], R[N];
    uint64_t lo = lds[q];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        uint64_t hi = lds[q + j + 1];
        W[j] = funnel64(lo, hi, s);
        lo = hi;
    }
    W[N - 1] &= mask;  // the head word of the kmer holds K*DST - 64(N-1) bits
#pragma unroll
    for (int j = 0; j < N; ++j) {
        rc[N - 1 - j] = comp_symbols<DST>(W[j]);
        R[j] = rev_symbols<DST>(W[j]);
    }
    if constexpr (DST == 2) rc[0] &= mask;  // NOT sets the unused top bits (transformations.jl:24)
    const uint32_t sh = 64u * N - (uint32_t)DST * k;  // 0..62
    fw[0] = R[0] >> sh;
#pragma unroll
    for (int i = 1; i < N; ++i) fw[i] = (R[i] >> sh) | ((R[i - 1] << 1) << (63u - sh));
}

// One-word kmers, two per lane: the window of kmer r and the symbol that follows it (which turns it
// into kmer r+1) both sit inside the same 128 stream bits, so one LDS read pair serves both.
template <int DST>
__device__ __forceinline__ void window1_and_next(const uint64_t *lds, uint32_t bit, uint32_t k, uint64_t mask,
                                                 uint64_t &fw, uint64_t &rc, uint64_t &next_sym) {
    const uint32_t q = bit >> 6, s = bit & 63u;
    const uint64_t lo = lds[q], hi = lds[q + 1];
    const uint64_t W0 = funnel64(lo, hi, s);                              // stream bits [s, s+64)
    const uint64_t W1 = (W0 >> DST) | ((hi >> s) << (64 - DST));          // stream bits [s+DST, s+DST+64)
    const uint64_t W = W0 & mask;
    rc = comp_symbols<DST>(W);
    if constexpr (DST == 2) rc &= mask;
    fw = rev_symbols<DST>(W) >> (64u - (uint32_t)DST * k);
    next_sym = (W1 >> ((uint32_t)DST * (k - 1u))) & ((1u << DST) - 1u);  // last symbol of the next window
}

// Kernels that only ever need FORWARD kmers of a 2-bit alphabet stage the codes of a tile in KMER order: symbol i of the staged
// words sits at bits [B - 2 - 2i, B - 2i) of the LDS stream (B = its length in bits), i.e. later symbols in LOWER bits,
// exactly like Kmer's big-endian layout (src/kmer.jl:32-44).  The kmer of the window whose first symbol is i is then the
// 2K bits at bit B - 2(i + K): one funnel shift and the head mask -- the same value as K applications of shift_encoding
// (construction_utils.jl:129-134), with no per-kmer symbol reversal.
__device__ __forceinline__ uint32_t rev2_32(uint32_t x) {  // reverse the order of the 16 two-bit symbols of a dword
    const uint32_t r = __brev(x);
    return ((r >> 1) & 0x55555555u) | ((r & 0x55555555u) << 1);
}
template <int N>
__device__ __forceinline__ void cut_fw(const uint64_t *rs, uint32_t o, uint64_t mask, uint64_t (&fw)[N]) {
    // the 64N stream bits from bit o on, read as 2N + 1 dwords from the dword that holds bit o: every output dword is one
    // v_alignbit_b32 of two neighbours (a 64-bit funnel shift is three 64-bit shifts and two ORs)
    const uint32_t *d = reinterpret_cast<const uint32_t *>(rs) + (o >> 5);
    const uint32_t b = o & 31u;
    uint32_t w[2 * N + 1];
#pragma unroll
    for (int i = 0; i < 2 * N + 1; ++i) w[i] = d[i];
#pragma unroll
    for (int j = 0; j < N; ++j) {  // word N-1-j of the kmer = stream bits [o + 64j, o + 64j + 64)
        const uint32_t lo = __builtin_amdgcn_alignbit(w[2 * j + 1], w[2 * j], b);
        const uint32_t hi = __builtin_amdgcn_alignbit(w[2 * j + 2], w[2 * j + 1], b);
        fw[N - 1 - j] = ((uint64_t)hi << 32) | lo;
    }
    fw[0] &= mask;
}

template <int N>
__device__ __forceinline__ bool kmer_less(const uint64_t (&x)[N], const uint64_t (&y)[N]) {
    // cmp(x.data, y.data) == -1: lexicographic, head word first (kmer.jl:176-178)
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (!decided && x[i] != y[i]) {
            lt = x[i] < y[i];
            decided = true;
        }
    }
    return lt;
}

template <int N>
__device__ __forceinline__ uint64_t fx_hash(const uint64_t (&x)[N], uint64_t seed) {
    uint64_t h = seed;
#pragma unroll
    for (int i = 0; i < N; ++i) h = fx_step(h, x[i]);
    return h;
}

// one kmer (N words) -> out[g*N ..]; 16-byte stores where the element size allows
template <int N>
__device__ __forceinline__ void store_kmer(uint64_t *out, uint64_t g, const uint64_t (&x)[N]) {
    if constexpr (N == 1) {
        out[g] = x[0];
    } else if constexpr (N == 2) {
        *reinterpret_cast<ulonglong2 *>(out + 2 * g) = make_ulonglong2(x[0], x[1]);
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int w = 0; w < N; w += 2)
            *reinterpret_cast<ulonglong2 *>(out + (uint64_t)N * g + w) = make_ulonglong2(x[w], x[w + 1]);
    } else {
#pragma unroll
        for (int w = 0; w < N; ++w) out[g * N + w] = x[w];
    }
}

// AsciiEncode of DNA / RNA text into a 2-bit alphabet without the table (text = 1: DNA, T valid; 2: RNA, U valid), 8 bytes at once:
// the 16 bits of their codes, and `off` != 0 iff one of them is not a symbol.  On the 32-bit halves, priced by
// profiles/r04_valu_rates.txt: the letter a code stands for comes from ONE v_perm (the code bytes select from "ACGT" / "ACGU"), the
// four codes of a half are gathered by ONE v_dot4 (bytes times 1, 4, 16, 64), and the word gets one verdict; the flag of every
// byte (text8_bad_bytes) is only computed if a byte is off.
__device__ __forceinline__ uint32_t text8_codes(uint64_t x, uint32_t text, uint32_t &off) {
    const uint32_t letters = text == 2u ? 0x55474341u : 0x54474341u;
    auto half = [&](uint32_t d, uint32_t &o) {
        const uint32_t U = d & 0xDFDFDFDFu;                   // upper case
        const uint32_t c2 = (U >> 1) & 0x03030303u;
        const uint32_t code = c2 ^ ((c2 >> 1) & 0x01010101u);
        o |= U ^ __builtin_amdgcn_perm(letters, letters, code);  // (selector bytes 0..3: bytes of `letters`)
        return __builtin_amdgcn_udot4(code, 0x40100401u, 0u, false);
    };
    off = 0;
    const uint32_t lo = half((uint32_t)x, off), hi = half((uint32_t)(x >> 32), off);
    return lo | (hi << 8);
}
// bit 8j: byte j of the word is not a symbol of the text's alphabet
__device__ __forceinline__ uint64_t text8_bad_bytes(uint64_t x, uint32_t text) {
    const uint64_t B1 = 0x0101010101010101ull;
    const uint64_t U = x & 0xDFDFDFDFDFDFDFDFull;
    const uint64_t c2 = (U >> 1) & (3ull * B1);
    const uint64_t code = c2 ^ ((c2 >> 1) & B1);
    const uint64_t b0 = code & B1, b1 = (code >> 1) & B1;
    const uint64_t is1 = b0 & ~b1, is2 = b1 & ~b0, is3 = b0 & b1;
    // the letter each code stands for: 'A' + {0, 2, 6, 0x13 ('T') or 0x14 ('U')}
    uint64_t E = 0x41ull * B1 + (is1 << 1) + (is2 << 1) + (is2 << 2) + (is3 << 4) + (is3 << 1) + is3;
    if (text == 2u) E += is3;
    const uint64_t bad = U ^ E;
    return ((bad | ((bad & (0x7Full * B1)) + (0x7Full * B1))) >> 7) & B1;
}

// Phase 1 for one source word: recode into the DST-bit LDS stream (RecodingScheme,
// src/construction.jl:75-100).  Returns one flag per offending symbol at bit SRC*j (FourToTwo:
// count_ones != 1; AsciiEncode: byte outside the alphabet), 0 otherwise.
//   stream word index: SRC == DST: qword wi;  4->2: dword wi;  2->4: qwords 2wi, 2wi+1
//   REV (2-bit kmers, forward only): the stream is kept in KMER order -- `wi` is then the word's index counted from the END of
//   the staged words and its symbols are reversed (cut_fw above)
template <int SRC, int DST, bool REV = false>
__device__ __forceinline__ uint64_t stage_word(uint64_t *lds, uint32_t wi, uint64_t x, const uint8_t *lut, uint32_t text = 0) {
    static_assert(!REV || DST == 2, "kmer-order staging: 2-bit kmer alphabets");
    if constexpr (SRC == 8 && DST == 2) {
        // AsciiEncode of DNA / RNA text into a 2-bit alphabet without the table (text = 1: DNA, T valid; 2: RNA, U valid): all
        // 8 bytes at once.  In either case of ACGT / ACGU, code = ((c >> 1) & 3) ^ (that >> 1) is A 0, C 1, G 2, T/U 3, and a
        // byte is a symbol iff its upper-cased value is the letter its code stands for (BioSequences.ascii_encode restated;
        // ascii_tables.hpp holds the table this replaces, tests/c/ascii_entry_check.cpp compares the two over all 256 bytes).
        if (text) {
            uint32_t off;
            const uint32_t codes = text8_codes(x, text, off);
            if constexpr (REV) reinterpret_cast<uint16_t *>(lds)[wi] = (uint16_t)(rev2_32(codes) >> 16);
            else reinterpret_cast<uint16_t *>(lds)[wi] = (uint16_t)codes;
            return off ? text8_bad_bytes(x, text) : 0;
        }
    }
    if constexpr (SRC == 8) {  // AsciiEncode: 8 bytes -> 8 symbols through the alphabet's table
        uint32_t codes = 0;
        uint64_t f = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            uint32_t v = lut[(x >> (8 * j)) & 0xffu];
            codes |= (v & 0xfu) << (DST * j);
            f |= (uint64_t)(v >> 7) << (8 * j);  // 0x80: not a symbol of the alphabet
        }
        if constexpr (REV) reinterpret_cast<uint16_t *>(lds)[wi] = (uint16_t)(rev2_32(codes) >> 16);
        else if constexpr (DST == 2) reinterpret_cast<uint16_t *>(lds)[wi] = (uint16_t)codes;
        else reinterpret_cast<uint32_t *>(lds)[wi] = codes;
        return f;
    } else if constexpr (SRC == DST) {  // Copyable
        lds[wi] = REV ? rev2(x) : x;
        return 0;
    } else if constexpr (SRC == 4) {  // FourToTwo: trailing_zeros of a one-hot nibble, validated
        uint32_t any_bad;
        const uint32_t c = pack_4to2_checked(x, any_bad);
        reinterpret_cast<uint32_t *>(lds)[wi] = REV ? rev2_32(c) : c;
        return any_bad ? flags_from_bad4(bad_nibbles4(x)) : 0;
    } else {  // TwoToFour
        lds[2 * wi] = expand_2to4((uint32_t)x);
        lds[2 * wi + 1] = expand_2to4((uint32_t)(x >> 32));
        return 0;
    }
}


// MinHash candidate h (already below the threshold): append it to out[] through the counter unless
// the same value was appended recently.  Low-complexity sequence (poly-A, tandem repeats) repeats
// a few hashes millions of times; without this filter they flood the candidate buffer.  `recent`
// is a direct-mapped table in HBM: a plain (L1-bypassing) load rejects repeats cheaply, the
// exchange makes "first to insert appends" exact; a collision merely evicts (the evicted value may
// be appended again later -- harmless, the sketch keeps distinct values).
__device__ __forceinline__ bool sketch_is_new(const StreamArgs &a, uint64_t h) {
    if (a.recent && h != ~0ull) {  // ~0 marks an empty slot
        unsigned long long *slot = reinterpret_cast<unsigned long long *>(a.recent) + ((uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 40) & a.recent_mask);
        if (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == h) return false;
        if (atomicExch(slot, (unsigned long long)h) == h) return false;
    }
    return true;
}
__device__ __forceinline__ void sketch_append(const StreamArgs &a, uint64_t h) {
    unsigned long long pos = atomicAdd(reinterpret_cast<unsigned long long *>(a.out_b), 1ull);
    if (pos < a.capacity) a.out_a[pos] = h;
}
__device__ __forceinline__ void sketch_candidate(const StreamArgs &a, uint64_t h) {
    if (sketch_is_new(a, h)) sketch_append(a, h);
}

// PAIR (strided kernels, one-word kmers): a lane takes two neighbouring lattice kmers, each cut from its own window, so
// that SpacedKmers output is stored 16 bytes per lane like the stride-1 kernels' (SpacedKmers.jl:121-139 yields the same
// elements whichever lane cuts them).
// FWD (MODE_FW over a 2-bit kmer alphabet with no reverse complements wanted: FwKmers, SpacedKmers, UnambiguousKmers over
// a sequence in which nothing is dropped): the codes are staged in kmer order, a kmer is one funnel shift + the head mask
// (cut_fw) -- no symbol reversal per kmer.  The strided kernel is bound by its integer work (one SIMD-cycle in four per
// instruction at 5.9 TB/s), so this is where it counts.
template <int SRC_BITS, int DST, int N, int MODE, bool STRIDE1, bool TUPLES = false, bool PAIR = false, bool FWD = false>
// (the strided, forward-only and text instances take 103-106 scalar registers = seven wavefronts per SIMD; compiled for eight
// -- amdgpu_waves_per_eu(8), what took ragged_kernel from seven tiles per CU to eight -- C5 strict went from 0.79 to 0.767 and f1
// from 0.807 to 0.813: not applied, round 6)
__global__ __launch_bounds__(BLOCK) void stream_kernel(const StreamArgs a) {
    static_assert(!FWD || (MODE == MODE_FW && DST == 2 && !TUPLES), "FWD: forward kmers of a 2-bit alphabet, separate arrays");
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    const uint32_t tid = threadIdx.x;
    // threads per workgroup: a LAUNCH parameter (64, 128 or 256: stream_launch.hpp picks it per output shape); the tile, not
    // the workgroup, fixes how much a workgroup writes
    const uint32_t TB = blockDim.x;
    const uint32_t k = a.k;
    const uint32_t J = STRIDE1 ? 1u : a.stride;
    const uint64_t mask = head_mask((int)k, DST);  // mask of the kmer's head word
    // byte sources: DNA / RNA text into a 2-bit alphabet is recoded arithmetically (stage_word), everything else through the
    // alphabet's 256-entry table in LDS
    const uint32_t text = (SRC_BITS == 8 && DST == 2 && a.ascii_table <= 1u) ? 1u + a.ascii_table : 0u;
    if constexpr (SRC_BITS == 8) {
        if (!text) {  // (uniform: every thread takes the same side)
            for (uint32_t i = tid; i < 256u; i += TB) lut[i] = ascii_entry(a.ascii_table, i);  // computed, not loaded
            // the strided kernels stage their first tile BEFORE the tile loop's first barrier: every wavefront's share of the table
            // must be in LDS before any wavefront looks a byte up (one barrier per workgroup, not per tile)
            block_sync();
        }
    }
    static_assert(!PAIR || (!STRIDE1 && N == 1 && !TUPLES && (MODE == MODE_FW || MODE == MODE_XOR)), "PAIR: strided one-word kmers");
    constexpr uint32_t KPL = ((STRIDE1 || PAIR) && N == 1 && !TUPLES) ? 2u : 1u;  // kmers per lane per pass -> 16 B stores
    uint64_t xacc = 0;
    uint64_t sketch_threshold = a.threshold;
    if constexpr (MODE == MODE_SKETCH) {
        if (a.threshold_ptr) sketch_threshold = *a.threshold_ptr;  // uniform; constant for the whole launch
    }
    // MODE_COUNT (K >= 11 only; smaller K use composition_kernel.hpp): plain global counters
    uint32_t *count_base = nullptr;
    if constexpr (MODE == MODE_COUNT) count_base = reinterpret_cast<uint32_t *>(a.out_a);

    // A strided tile reads `stride` times the source per output byte (769 words in front of the first store of 4096
    // SpacedDNAMers{21,3} windows), and under the kernel's own store traffic a round of loads takes several microseconds: a
    // workgroup of such a kernel visits `subtiles` CONSECUTIVE tiles and has the next tile's words in flight (registers) while it
    // stores the current one; the stream is double-buffered in LDS so that one barrier per tile suffices.  Stride-1 kernels keep
    // one tile per visit (their load round is a quarter of a load per lane; short-lived workgroups write fastest).
    constexpr uint32_t NBUF = STRIDE1 ? 1u : 2u;
    // words per lane of one load round (in flight together).  (Round 6 tried two for stride-1 tiles of TEXT, whose 128 x 1536 shape holds
    // 1.5 words per lane -- the one-round rule that took kmers_batch from text from 0.70 to 0.74: f1 stayed at 0.801-0.803.  These
    // tiles are short and sixteen to a CU; their second round hides behind the other tiles' stores.)
    constexpr uint32_t PRE = STRIDE1 ? 1u : (SRC_BITS == 8 ? 8u : 4u);
    __shared__ uint64_t lds_all[NBUF][LDS_QWORDS];
    struct Geom {
        uint64_t m0, w0;
        uint32_t mt, b0, nw, span;
    };
    auto geometry = [&](uint64_t tile) {
        Geom g;
        g.m0 = tile * a.tile_kmers;
        const uint64_t left = a.n_kmers - g.m0;
        g.mt = left < a.tile_kmers ? (uint32_t)left : a.tile_kmers;
        const uint64_t bit0 = a.first_bit + g.m0 * J * SRC_BITS;
        g.w0 = bit0 >> 6;
        g.b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        g.span = MODE == MODE_MINIMIZER ? k + a.window_kmers - 1u : k;  // symbols one element reads
        const uint64_t end_bit = bit0 + ((uint64_t)(g.mt - 1) * J + g.span) * SRC_BITS;
        g.nw = (uint32_t)(((end_bit + 63) >> 6) - g.w0);
        return g;
    };
    auto load_round = [&](const Geom &g, uint32_t base, uint64_t (&x)[PRE]) {  // all of a lane's loads of a round are issued together
#pragma unroll
        for (uint32_t j = 0; j < PRE; ++j) {
            const uint32_t wi = base + j * TB + tid;
            x[j] = wi < g.nw ? a.src[g.w0 + wi] : 0;
        }
    };
    auto stage_round = [&](const Geom &g, uint32_t base, const uint64_t (&x)[PRE], uint64_t *lds) {
#pragma unroll
        for (uint32_t j = 0; j < PRE; ++j) {
            const uint32_t wi = base + j * TB + tid;
            if (wi < g.nw) {
                uint64_t f = stage_word<SRC_BITS, DST, FWD>(lds, FWD ? g.nw - 1u - wi : wi, x[j], lut, text);
                if constexpr ((SRC_BITS == 4 && DST == 2) || SRC_BITS == 8) {
                    // `span` symbols per element are read (K, or K + W - 1 for minimizer windows): gaps start after them
                    if (f) report_bad_symbols<SRC_BITS, STRIDE1>(a.err_slot, a.first_bit, a.inspect_end, a.stride, g.span, g.w0 + wi, f, x[j], a.err_origin);
                }
            }
        }
    };
    const uint64_t S = STRIDE1 || a.subtiles == 0 ? 1u : a.subtiles;
    uint32_t buf = 0;
    // SPLIT ORDER (a.split_order; stream_launch.hpp decides): even workgroups walk the first half of the tile range, odd ones the
    // second half, so that every output array is written through TWO moving windows half an array apart.  On MI355X store streams
    // inside one region class of HBM share ~6 TB/s and streams in different classes reach ~7.2 (memory_api.hip,
    // profiles/r03_alloc.md): the ONE output array of a launch that lies across a class boundary (kmers_dev_alloc_role) is then
    // written 6-7 % faster at the launch shape that suits two windows (C3 0.81 -> 0.87, profiles/r03_tuning.md section 5; at the
    // one-window shape: nothing).  The two arrays of a two-output launch are better off in two different classes with one window
    // each (-1..4 % with four windows), which is what the context's arena arranges.
    const uint64_t n_visits = (a.n_tiles + S - 1) / S, n_first = (n_visits + 1) >> 1;
    const uint64_t n_slots = a.split_order ? 2 * n_first : n_visits;
    for (uint64_t slot = blockIdx.x; slot < n_slots; slot += gridDim.x) {
      uint64_t visit = slot;
      if (a.split_order) {
          visit = (slot & 1u) ? n_first + (slot >> 1) : (slot >> 1);
          if (visit >= n_visits) continue;  // (an odd number of visits: the second half is one short)
      }
      const uint64_t t_end = (visit + 1) * S < a.n_tiles ? (visit + 1) * S : a.n_tiles;
      uint64_t xpre[PRE];
      Geom gn = geometry(visit * S);
      if constexpr (!STRIDE1) load_round(gn, 0, xpre);  // the first tile of the visit: nothing to hide behind
      for (uint64_t tile = visit * S; tile < t_end; ++tile) {
        const Geom g = gn;
        const uint64_t m0 = g.m0;
        const uint32_t mt = g.mt, b0 = g.b0, nw = g.nw;
        uint64_t *const lds = lds_all[buf];
#ifdef KMERS_STAMPS
        uint64_t t0 = __builtin_amdgcn_s_memrealtime(), t1 = 0, t2 = 0, t3 = 0, t4 = 0;
#endif
        // ---- phase 1: source words -> DST-bit stream in LDS ------------------------------
        if constexpr (STRIDE1) {
            block_sync();  // previous tile's readers are done with the LDS stream
            for (uint32_t base = 0; base < nw; base += PRE * TB) {
                load_round(g, base, xpre);
                stage_round(g, base, xpre, lds);
            }
        } else {
            // (no barrier in front: this buffer's readers -- the tile before last -- all passed the barrier below since)
            stage_round(g, 0, xpre, lds);
            for (uint32_t base = PRE * TB; base < nw; base += PRE * TB) {  // (tiles longer than one round: not prefetched)
                load_round(g, base, xpre);
                stage_round(g, base, xpre, lds);
            }
        }
#ifdef KMERS_STAMPS
        t1 = __builtin_amdgcn_s_memrealtime();
#endif
        block_sync();
#ifdef KMERS_STAMPS
        t2 = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (!STRIDE1) {
            if (tile + 1 < t_end) {  // the next tile's words travel while this one is stored
                gn = geometry(tile + 1);
                load_round(gn, 0, xpre);
            }
            buf ^= 1u;
        }

        // ---- phase 2: windows -> kmers ---------------------------------------------------
        for (uint32_t r = tid * KPL; r < mt; r += TB * KPL) {
            const uint64_t g = m0 + r;
            uint64_t fw[KPL][N], rc[KPL][N];
            if constexpr (FWD) {
                // the kmer of tile element r sits at bit kbit0 - 2 J r of the kmer-order stream (rc stays unused)
                const uint32_t o = nw * (128u / SRC_BITS) - 2u * (b0 + k) - 2u * J * r;
#pragma unroll
                for (uint32_t e = 0; e < KPL; ++e)
#pragma unroll
                    for (int w = 0; w < N; ++w) rc[e][w] = 0;
                cut_fw<N>(lds, o, mask, fw[0]);
                if constexpr (KPL == 2) {
                    if (r + 1 < mt) cut_fw<N>(lds, o - 2u * J, mask, fw[1]);
                    else fw[1][0] = 0;
                }
            } else if constexpr (PAIR) {
                window<N, DST>(lds, (uint32_t)DST * (r * J + b0), k, mask, fw[0], rc[0]);
                if (r + 1 < mt) window<N, DST>(lds, (uint32_t)DST * ((r + 1) * J + b0), k, mask, fw[1], rc[1]);
                else fw[1][0] = rc[1][0] = 0;
            } else if constexpr (KPL == 2) {
                // next window: one symbol further.  fw shifts left, rc shifts right with the
                // complemented symbol on top (the reference's own rolling step,
                // CanonicalKmers.jl:102-103, :115-118).
                uint64_t sym;
                window1_and_next<DST>(lds, (uint32_t)DST * (r + b0), k, mask, fw[0][0], rc[0][0], sym);
                uint64_t csym;
                if constexpr (DST == 2) csym = sym ^ 3u;
                else csym = ((sym & 1u) << 3) | ((sym & 2u) << 1) | ((sym & 4u) >> 1) | ((sym & 8u) >> 3);
                fw[1][0] = ((fw[0][0] << DST) | sym) & mask;
                rc[1][0] = (rc[0][0] >> DST) | (csym << ((uint32_t)DST * (k - 1u)));
            } else {
                window<N, DST>(lds, (uint32_t)DST * (r * J + b0), k, mask, fw[0], rc[0]);
            }
            const bool both = (KPL == 2) && (r + 1 < mt);

            if constexpr (MODE == MODE_MINIMIZER) {
                // docs/src/replacements.md:33-51: start from the window's first kmer, shift the next
                // W-1 symbols in one at a time, keep the kmer with the smallest fx_hash
                uint64_t best[N], cur[N];
#pragma unroll
                for (int w = 0; w < N; ++w) best[w] = cur[w] = fw[0][w];
                uint64_t hash = fx_hash<N>(best, 0);
                for (uint32_t off = 0; off + 1 < a.window_kmers; ++off) {
                    const uint32_t bit = (uint32_t)DST * (r * J + b0 + k + off);
                    const uint64_t sym = (lds[bit >> 6] >> (bit & 63u)) & ((1u << DST) - 1u);
                    uint64_t nk[N];
                    // mode 0 shifts into the CURRENT MINIMUM (as the published example does), mode 1 into the rolling kmer
#pragma unroll
                    for (int w = 0; w < N; ++w) nk[w] = a.minimizer_mode == 0 ? best[w] : cur[w];
#pragma unroll
                    for (int w = 0; w < N - 1; ++w) nk[w] = (nk[w] << DST) | (nk[w + 1] >> (64 - DST));  // shift_encoding
                    nk[N - 1] = (nk[N - 1] << DST) | sym;
                    nk[0] &= mask;
#pragma unroll
                    for (int w = 0; w < N; ++w) cur[w] = nk[w];
                    const uint64_t nh = fx_hash<N>(nk, 0);
                    if (nh < hash) {
                        hash = nh;
#pragma unroll
                        for (int w = 0; w < N; ++w) best[w] = nk[w];
                    }
                }
                store_kmer<N>(a.out_a, g, best);
            } else if constexpr (MODE == MODE_FW && TUPLES) {
                // Tuple{Kmer,Kmer} elements of FwRvIterator (CanonicalKmers.jl:44-45): fw words, then rc words
                uint64_t t[2 * N];
#pragma unroll
                for (int w = 0; w < N; ++w) {
                    t[w] = fw[0][w];
                    t[N + w] = rc[0][w];
                }
                store_kmer<2 * N>(a.out_a, g, t);
            } else if constexpr (MODE == MODE_FW) {
                const uint64_t start = g + 1 + a.start_origin;
                if constexpr (KPL == 2) {
                    if (both) {
                        if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(fw[0][0], fw[1][0]);
                        if (a.out_b) *reinterpret_cast<ulonglong2 *>(a.out_b + g) = make_ulonglong2(rc[0][0], rc[1][0]);
                        if (a.out_starts) *reinterpret_cast<ulonglong2 *>(a.out_starts + g) = make_ulonglong2(start, start + 1);
                    } else {
                        if (a.out_a) a.out_a[g] = fw[0][0];
                        if (a.out_b) a.out_b[g] = rc[0][0];
                        if (a.out_starts) a.out_starts[g] = (long long)start;
                    }
                } else {
                    if (a.out_a) store_kmer<N>(a.out_a, g, fw[0]);
                    if (a.out_b) store_kmer<N>(a.out_b, g, rc[0]);
                    if (a.out_starts) a.out_starts[g] = (long long)start;
                }
            } else {
                // canonical: fw < rv ? fw : rv (CanonicalKmers.jl:224)
                uint64_t c[KPL][N];
#pragma unroll
                for (uint32_t e = 0; e < KPL; ++e) {
                    bool lt = kmer_less<N>(fw[e], rc[e]);
#pragma unroll
                    for (int w = 0; w < N; ++w) c[e][w] = lt ? fw[e][w] : rc[e][w];
                }
                if constexpr (MODE == MODE_XOR) {
                    const bool can = a.xor_canonical != 0;
                    xacc ^= can ? c[0][0] : fw[0][0];
                    if constexpr (KPL == 2) {
                        if (both) xacc ^= can ? c[1][0] : fw[1][0];
                    }
                } else if constexpr (MODE == MODE_SKETCH) {
                    // bottom-s MinHash candidates: fx_hash(canonical kmer) below the running threshold
#pragma unroll
                    for (uint32_t e = 0; e < KPL; ++e) {
                        if (e == 0 || both) {
                            const uint64_t h = fx_hash<N>(c[e], a.seed);
                            if (h < sketch_threshold) sketch_candidate(a, h);
                        }
                    }
                } else if constexpr (MODE == MODE_COUNT) {
                    // kmer composition: counts[as_integer(kmer)] += 1 (docs/src/composition.md:31-33)
                    atomicAdd(count_base + (uint32_t)fw[0][0], 1u);
                    if constexpr (KPL == 2) {
                        if (both) atomicAdd(count_base + (uint32_t)fw[1][0], 1u);
                    }
                } else if constexpr (TUPLES) {
                    // Tuple{Kmer,UInt64} elements: the canonical kmer, then its fx_hash
                    uint64_t t[N + 1];
#pragma unroll
                    for (int w = 0; w < N; ++w) t[w] = c[0][w];
                    t[N] = fx_hash<N>(c[0], a.seed);
                    store_kmer<N + 1>(a.out_a, g, t);
                } else if constexpr (KPL == 2) {
                    if (both) {
                        if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(c[0][0], c[1][0]);
                        if (a.out_b)
                            *reinterpret_cast<ulonglong2 *>(a.out_b + g) =
                                make_ulonglong2(fx_hash<N>(c[0], a.seed), fx_hash<N>(c[1], a.seed));
                    } else {
                        if (a.out_a) a.out_a[g] = c[0][0];
                        if (a.out_b) a.out_b[g] = fx_hash<N>(c[0], a.seed);
                    }
                } else {
                    if (a.out_a) store_kmer<N>(a.out_a, g, c[0]);
                    if (a.out_b) a.out_b[g] = fx_hash<N>(c[0], a.seed);
                }
            }
        }
#ifdef KMERS_STAMPS
        t3 = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        t4 = __builtin_amdgcn_s_memrealtime();
        if (a.stamps && (tile & 1023u) == 0 && (tid & 63u) == 0) {
            uint64_t *o = a.stamps + ((tile >> 10) * (TB >> 6) + (tid >> 6)) * 8;
            o[0] = t0; o[1] = t1; o[2] = t2; o[3] = t3; o[4] = t4; o[5] = tile;
        }
#endif
      }
    }

    if constexpr (MODE == MODE_XOR) {
        // wavefront XOR-reduce (64 lanes), then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if ((tid & 63u) == 0) atomicXor(reinterpret_cast<unsigned long long *>(a.out_a), (unsigned long long)xacc);
    }
}

// Direct (gather) kernel for large strides, where a tile would stage mostly unused symbols:
// one lane per kmer, symbols fetched and shifted in one by one exactly like unsafe_extract
// (src/construction_utils.jl:27-69).  Edge path, not bandwidth critical.
template <int SRC_BITS, int DST, int N>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const StreamArgs a) {
    uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n_kmers) return;
    uint64_t base = g * a.stride;
    uint64_t d[N];
#pragma unroll
    for (int w = 0; w < N; ++w) d[w] = 0;
    for (uint32_t t = 0; t < a.k; ++t) {
        uint64_t bit = a.first_bit + (base + t) * SRC_BITS;
        uint64_t enc = (a.src[bit >> 6] >> (bit & 63u)) & ((1ull << SRC_BITS) - 1ull);
        uint64_t code = enc;
        if constexpr (SRC_BITS == 8) {
            code = ascii_entry(a.ascii_table, (uint32_t)enc);
            if (code & 0x80u) {
                atomicMin(a.err_slot, error_key(base + t + a.err_origin, enc));
                return;
            }
        } else if constexpr (SRC_BITS == 4 && DST == 2) {
            if (__popcll(enc) != 1) {
                atomicMin(a.err_slot, error_key(base + t + a.err_origin, enc));
                return;
            }
            code = (uint64_t)(__ffsll((long long)enc) - 1);
        } else if constexpr (SRC_BITS == 2 && DST == 4) {
            code = 1ull << enc;
        }
#pragma unroll
        for (int w = 0; w < N - 1; ++w) d[w] = (d[w] << DST) | (d[w + 1] >> (64 - DST));  // leftshift_carry
        d[N - 1] = (d[N - 1] << DST) | code;
    }
    store_kmer<N>(a.out_a, g, d);
}

}  // namespace kmers
