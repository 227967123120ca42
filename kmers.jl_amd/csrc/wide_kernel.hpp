// wide_kernel.hpp -- kmers of ANY width: Kmer{A,K,N} has no upper bound on N in the reference (src/kmer.jl:97-111:
// N = cld(K * bits_per_symbol, 64)); the tile kernels are compiled for N = 1..4.  This kernel takes N at run time, so that
// no reference-legal (A, K) over 2-bit / 4-bit / byte sources is refused by FwKmers, FwRvIterator, CanonicalKmers
// (+ fx_hash) and SpacedKmers (src/iterators/FwKmers.jl:57-115, CanonicalKmers.jl:54-144, :220-225, SpacedKmers.jl:83-139,
// src/kmer.jl:255-261).  One lane per kmer; a word of the kmer is assembled straight from the source symbols it holds
// (64 / bits of them), so nothing is indexed dynamically and no register array depends on N.  An edge path (2 KiB
// kmers are not a throughput workload): scattered 8-byte reads through L1/L2, not tuned.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

// recoded symbol `idx` (0-based inside the view) per RecodingScheme (src/construction.jl:75-100); a symbol the kmer
// alphabet cannot hold reports the reference's EncodeError and reads as 0
template <int SRC_BITS, int DST>
__device__ __forceinline__ uint64_t wide_symbol(const StreamArgs &a, uint64_t idx) {
    const uint64_t bit = a.first_bit + idx * SRC_BITS;
    const uint64_t enc = (a.src[bit >> 6] >> (bit & 63u)) & ((1ull << SRC_BITS) - 1ull);
    if constexpr (SRC_BITS == 8) {
        const uint32_t code = ascii_entry(a.ascii_table, (uint32_t)enc);
        if (code & 0x80u) {
            atomicMin(a.err_slot, error_key(idx + a.err_origin, enc));
            return 0;
        }
        return code;
    } else if constexpr (SRC_BITS == 4 && DST == 2) {
        if (__popcll(enc) != 1) {  // construction_utils.jl:50
            atomicMin(a.err_slot, error_key(idx + a.err_origin, enc));
            return 0;
        }
        return (uint64_t)(__ffsll((long long)enc) - 1);
    } else if constexpr (SRC_BITS == 2 && DST == 4) {
        return 1ull << enc;
    } else {
        return enc;
    }
}

template <int DST>
__device__ __forceinline__ uint64_t comp_one(uint64_t s) {
    if constexpr (DST == 2) return s ^ 3ull;
    return ((s & 1u) << 3) | ((s & 2u) << 1) | ((s & 4u) >> 1) | ((s & 8u) >> 3);
}

// Word w (0 = head, data[1] in Julia) of the forward kmer of the window starting at symbol `start`, or of its reverse
// complement.  Slot p of a kmer (p = 0 its first symbol) sits at bit DST * (K - 1 - p) of the N-word integer; word w holds
// the slots whose bits fall into [64 (N-1-w), 64 (N-w)).  Slot p of the reverse complement is the complement of the
// window's symbol K - 1 - p (transformations.jl:1-34).
template <int SRC_BITS, int DST>
__device__ __forceinline__ uint64_t wide_word(const StreamArgs &a, uint64_t start, uint32_t n_words, uint32_t w, bool rc) {
    const long long K = (long long)a.k;
    const long long top = 64ll * (long long)(n_words - w);         // exclusive upper bit of the word
    long long p_lo = K - 1 - (top - DST) / DST;                     // slot in the word's highest symbol position ...
    if (p_lo < 0) p_lo = 0;                                         // ... (the head word holds fewer: bits_unused, kmer.jl:128)
    const long long p_hi = K - 1 - (top - 64) / DST;                // slot in its lowest
    uint64_t word = 0;
    for (long long p = p_lo; p <= p_hi; ++p) {
        const uint64_t s = rc ? comp_one<DST>(wide_symbol<SRC_BITS, DST>(a, start + (uint64_t)(K - 1 - p)))
                              : wide_symbol<SRC_BITS, DST>(a, start + (uint64_t)p);
        word = (word << DST) | s;
    }
    return word;
}

// MODE_FW: out_a = forward kmers, out_b = reverse complements (nullable).  MODE_CANON: out_a = canonical kmers
// (nullable), out_b = fx_hash(canonical kmer, seed) (nullable).
template <int SRC_BITS, int DST, int MODE>
__global__ __launch_bounds__(BLOCK) void wide_kernel(const StreamArgs a, const uint32_t n_words) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n_kmers) return;
    const uint64_t start = g * a.stride;
    if constexpr (MODE == MODE_FW) {
        for (uint32_t w = 0; w < n_words; ++w) {
            a.out_a[g * n_words + w] = wide_word<SRC_BITS, DST>(a, start, n_words, w, false);
            if (a.out_b) a.out_b[g * n_words + w] = wide_word<SRC_BITS, DST>(a, start, n_words, w, true);
        }
    } else {
        // fw < rv ? fw : rv (CanonicalKmers.jl:224): lexicographic on the word tuples, head first (kmer.jl:176-178)
        bool take_fw = false;
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t f = wide_word<SRC_BITS, DST>(a, start, n_words, w, false), r = wide_word<SRC_BITS, DST>(a, start, n_words, w, true);
            if (f != r) {
                take_fw = f < r;
                break;
            }
        }
        uint64_t h = a.seed;
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t c = wide_word<SRC_BITS, DST>(a, start, n_words, w, !take_fw);
            if (a.out_a) a.out_a[g * n_words + w] = c;
            h = fx_step(h, c);  // kmer.jl:255-260
        }
        if (a.out_b) a.out_b[g] = h;
    }
}

}  // namespace kmers
