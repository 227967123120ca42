// device_bits.hpp -- word-level primitives shared by the gfx950 kernels.
//
// Everything here is unsigned 64-bit integer arithmetic (no floating point on this
// path).  The kernels never roll a kmer symbol by symbol as the reference does
// (src/iterators/CanonicalKmers.jl:131-144); they cut each window out of a 2-bit
// little-endian stream staged in LDS and derive both strands from it:
//   W            = bits [2i, 2i+2K) of the stream (base j of the window at bits 2j)
//   rc kmer      = ~W & mask                (complement = NOT, order already reversed)
//   forward kmer = rev2(W) >> (64N - 2K)    (symbol order reversed, big-endian Kmer layout)
// which is bit-identical to shift_encoding / shift_first_encoding (construction_utils.jl:129-134,
// kmer.jl:511-518) applied K times, see DESIGN.md.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kmers {

constexpr uint64_t FX_CONSTANT = 0x517cc1b727220a95ull;  // src/kmer.jl:218
constexpr uint64_t NO_ERROR_POS = ~0ull;

// ---- geometry (host+device), src/kmer.jl:117-137, :603-605 ------------------------------
__host__ __device__ inline int n_coding_elements(int k, int bps) { return (k * bps + 63) / 64; }
__host__ __device__ inline int bits_unused(int k, int bps) {
    return (64 / bps * n_coding_elements(k, bps) - k) * bps;
}
// get_mask: all ones when bits_unused == 0 (Julia's 1 << 64 == 0; in C that shift is UB)
__host__ __device__ inline uint64_t head_mask(int k, int bps) {
    int bu = bits_unused(k, bps);
    return bu == 0 ? ~0ull : ((1ull << (64 - bu)) - 1ull);
}

// ---- 4-bit one-hot -> 2-bit codes, 16 symbols of one LongDNA{4} word at a time ------------
// code = trailing_zeros(nibble) (construction_utils.jl:51): bit0 = C|T, bit1 = G|T.
// `bad` gets a non-zero nibble for every symbol with count_ones != 1 (construction_utils.jl:50).
// (The plain statement of the recoding, kept for reference: since round 4 the kernels call pack_4to2_checked below.)
__device__ __forceinline__ uint32_t pack_4to2(uint64_t x, uint64_t &bad) {
    const uint64_t M1 = 0x1111111111111111ull;
    uint64_t x1 = x >> 1, x2 = x >> 2, x3 = x >> 3;
    uint64_t pop = (x & M1) + (x1 & M1) + (x2 & M1) + (x3 & M1);
    bad = pop ^ M1;
    const uint64_t c = ((x1 | x3) & M1) | (((x2 | x3) & M1) << 1);  // 2-bit code in the low bits of each nibble
    // 16 nibbles -> 16 2-bit fields, on the 32-bit halves: pairs of nibbles into the low nibble of every byte, pairs of
    // bytes into bytes 0 and 2, and one byte permute gathers the four bytes of the two halves (11 instructions; the
    // 64-bit shift-or-mask ladder was twice that, and these paths are bound by their instruction count)
    uint32_t lo = (uint32_t)c, hi = (uint32_t)(c >> 32);
    lo = (lo | (lo >> 2)) & 0x0F0F0F0Fu;
    hi = (hi | (hi >> 2)) & 0x0F0F0F0Fu;
    lo |= lo >> 4;
    hi |= hi >> 4;
    return __builtin_amdgcn_perm(hi, lo, 0x06040200u);  // bytes: lo.0, lo.2, hi.0, hi.2
}

// The same codes at what the SIMDs charge (profiles/r04_valu_rates.txt: 2.24 cycles for a simple two-operand integer instruction,
// 4.1 for every other one -- a 64-bit shift, v_perm, v_bcnt): on the 32-bit halves, trailing_zeros of a one-hot nibble n as
// (n >> 1) - (n >> 3) (1, 2, 4, 8 -> 0, 1, 2, 3; no borrow leaves a nibble), and ONE verdict for the word instead of a flag per
// symbol: every nibble non-zero and 16 bits set <=> every symbol one-hot.  About 85 cycles of a SIMD per word where pack_4to2
// takes about 300; the caller computes pack_4to2's `bad` only if `any_bad` says so.  Codes of symbols that are not one-hot are
// unspecified (the call fails on them).
__device__ __forceinline__ uint32_t pack_4to2_checked(uint64_t x, uint32_t &any_bad) {
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    auto half = [](uint32_t h) {
        // 2-bit code in the low bits of each nibble; a nibble that is NOT one-hot (a symbol outside the view or past the end that
        // shares the word: nobody asks for its code) must not spill into its neighbour's field below: 0xF gives 7 - 1 = 6
        const uint32_t c = (((h >> 1) & 0x77777777u) - ((h >> 3) & 0x11111111u)) & 0x33333333u;
        const uint32_t u = (c | (c >> 2)) & 0x0F0F0F0Fu;
        return u | (u >> 4);  // bytes 0 and 2: four codes each
    };
    const uint32_t nz_lo = ((lo & 0x77777777u) + 0x77777777u) | lo, nz_hi = ((hi & 0x77777777u) + 0x77777777u) | hi;  // bit 3 of a nibble: it is not 0
    any_bad = (~(nz_lo & nz_hi) & 0x88888888u) | ((uint32_t)__popcll(x) ^ 16u);
    return __builtin_amdgcn_perm(half(hi), half(lo), 0x06040200u);  // bytes: lo.0, lo.2, hi.0, hi.2
}
// pack_4to2's `bad` alone (the rare path behind pack_4to2_checked)
__device__ __forceinline__ uint64_t bad_nibbles4(uint64_t x) {
    const uint64_t M1 = 0x1111111111111111ull;
    return ((x & M1) + ((x >> 1) & M1) + ((x >> 2) & M1) + ((x >> 3) & M1)) ^ M1;
}

// one bit per symbol (bit j set = symbol j of the word is ambiguous) from pack_4to2's `bad`
__device__ __forceinline__ uint32_t bad_bits16(uint64_t bad) {
    uint64_t f = (bad | (bad >> 1) | (bad >> 2)) & 0x1111111111111111ull;
    f = (f | (f >> 3)) & 0x0303030303030303ull;
    f = (f | (f >> 6)) & 0x000F000F000F000Full;
    f = (f | (f >> 12)) & 0x000000FF000000FFull;
    f = (f | (f >> 24));
    return (uint32_t)f & 0xFFFFu;
}

// The same recoding for the kernels that keep their LDS stream in KMER order (cut_fw: symbol 15 of the word in bits [0, 2),
// symbol 0 in bits [30, 32)) and want one ambiguity bit per symbol: done on the 32-bit halves, the symbol reversal folded into
// one v_bfrev per half (after it A = 8, C = 4, G = 2, T = 1: code bit 0 = nibble bits 2 | 0, bit 1 = nibble bits 1 | 0).  41
// vector instructions per word where pack_4to2 + rev2_32 + bad_bits16 take 59 (profiles/r04_unamb.md).  The code of a symbol
// that is not one-hot is unspecified (its windows are never kept).
__device__ __forceinline__ uint32_t recode4_half_kmer_order(uint32_t r) {  // r = __brev(half word): 8 symbols, last one first
    const uint32_t c = ((r | (r >> 2)) & 0x11111111u) | ((r | (r << 1)) & 0x22222222u);  // 2-bit code in the low bits of each nibble
    const uint32_t u = (c | (c >> 2)) & 0x0F0F0F0Fu;
    return u | (u >> 4);  // bytes 0 and 2: four codes each
}
__device__ __forceinline__ uint32_t recode4_kmer_order(uint64_t x) {
    const uint32_t vl = recode4_half_kmer_order(__brev((uint32_t)x)), vh = recode4_half_kmer_order(__brev((uint32_t)(x >> 32)));
    return __builtin_amdgcn_perm(vl, vh, 0x06040200u);  // bytes: vh.0, vh.2 (symbols 15..8), vl.0, vl.2 (symbols 7..0)
}
// bit j set: symbol j of the word has count_ones != 1 (construction_utils.jl:50).  With p0 = n0 | n1, p1 = n2 | n3, q0 = n0 & n1,
// q1 = n2 & n3 a nibble is one-hot iff (p0 ^ p1) & ~q0 & ~q1; the eight flags of a half word (one per nibble) are gathered by
// one v_dot4_u32_u8 (bytes of two 2-bit fields times 1, 4, 16, 64).
__device__ __forceinline__ uint32_t ambiguous8(uint32_t h) {
    const uint32_t s = h >> 1, o = h | s, n = h & s;
    const uint32_t f = (~(o ^ (o >> 2)) | n | (n >> 2)) & 0x11111111u;
    const uint32_t g = (f | (f >> 3)) & 0x03030303u;
    return __builtin_amdgcn_udot4(g, 0x40100401u, 0u, false);
}
__device__ __forceinline__ uint32_t ambiguous16(uint64_t x) {
    return ambiguous8((uint32_t)x) | (ambiguous8((uint32_t)(x >> 32)) << 8);
}

// Put this before the __syncthreads() that follows LDS atomics WITHOUT a return value (ds_add_u32, ds_max_u32 ...)
// whose results other wavefronts read after the barrier.  hipcc emits a bare s_barrier there: its workgroup-scope
// release relies on the LDS queue being in order, and under heavy same-address contention that was observed to be
// false (a histogram word read after the barrier missed a whole tile of increments: the 16-bit composition counters
// overflowed in 1-2 % of cold runs).  The explicit wait makes every wavefront's own LDS operations complete first.
#ifdef KMERS_NO_SETTLE  // ISA comparison builds only (tools/check_lds_barrier.sh): the barriers as hipcc emits them
__device__ __forceinline__ void lds_atomics_settle() {}
#else
__device__ __forceinline__ void lds_atomics_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
// Every workgroup barrier of these kernels: the wavefront's own LDS operations first, then the barrier (the wait
// is what hipcc emits in most places anyway; making it unconditional removes the class of hazards above).
__device__ __forceinline__ void block_sync() {
    lds_atomics_settle();
    __syncthreads();
}

// reverse the order of the 32 two-bit symbols of a word (BioSequences.reversebits, bps = 2)
__device__ __forceinline__ uint64_t rev2(uint64_t x) {
    uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
}

// 64 stream bits starting at bit s (0..63) of the pair (lo, hi)
__device__ __forceinline__ uint64_t funnel64(uint64_t lo, uint64_t hi, uint32_t s) {
    return (lo >> s) | ((hi << 1) << (63u - s));
}

// fx_hash step, src/kmer.jl:255-260
__device__ __forceinline__ uint64_t fx_step(uint64_t h, uint64_t w) {
    return (((h << 5) | (h >> 59)) ^ w) * FX_CONSTANT;
}

// SplitMix64 finaliser: the build's synthetic generator
__host__ __device__ inline uint64_t synth_rand64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace kmers
