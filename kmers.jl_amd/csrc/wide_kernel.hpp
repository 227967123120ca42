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

// Word w (0 = head, data[1] in Julia) of the forward kmer of a window, or of its reverse complement; sym(p) = the
// recoded symbol p (0-based) of the window.  Slot p of a kmer (p = 0 its first symbol) sits at bit DST * (K - 1 - p) of the
// N-word integer; word w holds the slots whose bits fall into [64 (N-1-w), 64 (N-w)).  Slot p of the reverse complement is
// the complement of the window's symbol K - 1 - p (transformations.jl:1-34).
template <int DST, class Sym>
__device__ __forceinline__ uint64_t wide_word_of(Sym sym, uint32_t k, uint32_t n_words, uint32_t w, bool rc) {
    const long long K = (long long)k;
    const long long top = 64ll * (long long)(n_words - w);         // exclusive upper bit of the word
    long long p_lo = K - 1 - (top - DST) / DST;                     // slot in the word's highest symbol position ...
    if (p_lo < 0) p_lo = 0;                                         // ... (the head word holds fewer: bits_unused, kmer.jl:128)
    const long long p_hi = K - 1 - (top - 64) / DST;                // slot in its lowest
    uint64_t word = 0;
    for (long long p = p_lo; p <= p_hi; ++p) {
        const uint64_t s = rc ? comp_one<DST>(sym((uint64_t)(K - 1 - p))) : sym((uint64_t)p);
        word = (word << DST) | s;
    }
    return word;
}

// fw < rv ? fw : rv (CanonicalKmers.jl:224): lexicographic on the word tuples, head first (kmer.jl:176-178)
template <int DST, class Sym>
__device__ __forceinline__ bool wide_forward_is_canonical_of(Sym sym, uint32_t k, uint32_t n_words) {
    for (uint32_t w = 0; w < n_words; ++w) {
        const uint64_t f = wide_word_of<DST>(sym, k, n_words, w, false), r = wide_word_of<DST>(sym, k, n_words, w, true);
        if (f != r) return f < r;
    }
    return false;
}

// The same two words cut out of a stream that already holds the kmer alphabet's symbols, DST bits each, little-endian
// (symbol j at bits DST * j: a recoded pool in HBM, a staged tile in LDS): word w of the forward kmer of the window that
// starts at stream symbol p is 64 / DST consecutive symbols reversed, of its reverse complement the complement of the mirrored
// stretch -- two stream words and a funnel shift instead of 64 / DST single symbols.  load(q) = stream word q; the word behind
// a chunk's last symbol is never read.
template <int DST, class Load>
__device__ __forceinline__ uint64_t wide_word_from_stream(Load load, uint64_t p, uint32_t k, uint32_t n_words, uint32_t w, bool rc) {
    constexpr uint32_t SPW = 64u / (uint32_t)DST;
    const uint32_t c = w == 0u ? k - (n_words - 1u) * SPW : SPW;    // symbols in the word (the head word holds fewer, kmer.jl:128)
    const uint32_t p_lo = w == 0u ? 0u : k - (n_words - w) * SPW;   // the window's symbol in the word's top position
    const uint64_t b = (p + (rc ? k - p_lo - c : p_lo)) * (uint64_t)DST, q = b >> 6;
    const uint32_t sh = (uint32_t)(b & 63u);
    uint64_t v = funnel64(load(q), sh + c * (uint32_t)DST > 64u ? load(q + 1u) : 0ull, sh);
    const uint64_t keep = c < SPW ? (1ull << (c * (uint32_t)DST)) - 1ull : ~0ull;
    v &= keep;
    if (!rc) return rev_symbols<DST>(v) >> (64u - c * (uint32_t)DST);
    return comp_symbols<DST>(v) & keep;
}

template <int DST, class Load>
__device__ __forceinline__ bool wide_forward_is_canonical_from_stream(Load load, uint64_t p, uint32_t k, uint32_t n_words) {
    for (uint32_t w = 0; w < n_words; ++w) {
        const uint64_t f = wide_word_from_stream<DST>(load, p, k, n_words, w, false), r = wide_word_from_stream<DST>(load, p, k, n_words, w, true);
        if (f != r) return f < r;
    }
    return false;
}

// the same over a sequence view: the window that starts at symbol `start`
template <int SRC_BITS, int DST>
__device__ __forceinline__ uint64_t wide_word(const StreamArgs &a, uint64_t start, uint32_t n_words, uint32_t w, bool rc) {
    return wide_word_of<DST>([&](uint64_t p) { return wide_symbol<SRC_BITS, DST>(a, start + p); }, a.k, n_words, w, rc);
}

template <int SRC_BITS, int DST>
__device__ __forceinline__ bool wide_forward_is_canonical(const StreamArgs &a, uint64_t start, uint32_t n_words) {
    return wide_forward_is_canonical_of<DST>([&](uint64_t p) { return wide_symbol<SRC_BITS, DST>(a, start + p); }, a.k, n_words);
}

// MODE_FW: out_a = forward kmers, out_b = reverse complements (nullable).  MODE_CANON: out_a = canonical kmers
// (nullable), out_b = fx_hash(canonical kmer, seed) (nullable).  a.tuples: one array of Tuple{Kmer,Kmer} /
// Tuple{Kmer,UInt64} elements in out_a (CanonicalKmers.jl:44-45).
template <int SRC_BITS, int DST, int MODE>
__global__ __launch_bounds__(BLOCK) void wide_kernel(const StreamArgs a, const uint32_t n_words) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n_kmers) return;
    const uint64_t start = g * a.stride;
    if constexpr (MODE == MODE_FW) {
        uint64_t *fw = a.tuples ? a.out_a + g * 2u * n_words : (a.out_a ? a.out_a + g * n_words : nullptr);
        uint64_t *rc = a.tuples ? fw + n_words : (a.out_b ? a.out_b + g * n_words : nullptr);
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t f = wide_word<SRC_BITS, DST>(a, start, n_words, w, false);  // (computed even if not stored: it validates the symbols)
            if (fw) fw[w] = f;
            if (rc) rc[w] = wide_word<SRC_BITS, DST>(a, start, n_words, w, true);
        }
    } else {
        const bool take_fw = wide_forward_is_canonical<SRC_BITS, DST>(a, start, n_words);
        uint64_t *kmer = a.tuples ? a.out_a + g * (n_words + 1u) : (a.out_a ? a.out_a + g * n_words : nullptr);
        uint64_t h = a.seed;
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t c = wide_word<SRC_BITS, DST>(a, start, n_words, w, !take_fw);
            if (kmer) kmer[w] = c;
            h = fx_step(h, c);  // kmer.jl:255-260
        }
        if (a.tuples) kmer[n_words] = h;
        else if (a.out_b) a.out_b[g] = h;
    }
}

// ---- the fused consumers over kmers of any width (consumers_api.hip) ---------------------------------------------
// WIDE_XOR: the reducer of test/benchmark.jl:9-15 (XOR of data[1]) over FwKmers / CanonicalKmers / SpacedKmers;
// WIDE_SKETCH: bottom-s MinHash candidates, fx_hash(canonical kmer) below the running threshold (docs/src/minhash.md:17-41).
// A lane reads every symbol of its window (all words are computed), so every symbol the reference's loop would have
// inspected is validated.
enum WideConsumer { WIDE_XOR = 0, WIDE_SKETCH = 1 };
template <int SRC_BITS, int DST, int CMODE>
__global__ __launch_bounds__(BLOCK) void wide_consumer_kernel(const StreamArgs a, const uint32_t n_words) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    uint64_t xacc = 0;
    if (g < a.n_kmers) {
        const uint64_t start = g * a.stride;
        const bool canonical = CMODE == WIDE_SKETCH || a.xor_canonical != 0;
        const bool take_fw = canonical ? wide_forward_is_canonical<SRC_BITS, DST>(a, start, n_words) : true;
        uint64_t h = a.seed, head = 0;
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t c = wide_word<SRC_BITS, DST>(a, start, n_words, w, !take_fw);
            if (w == 0) head = c;
            h = fx_step(h, c);
        }
        if constexpr (CMODE == WIDE_XOR) {
            xacc = head;
        } else {
            const uint64_t threshold = a.threshold_ptr ? __hip_atomic_load(a.threshold_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.threshold;
            if (h < threshold) sketch_candidate(a, h);
        }
    }
    if constexpr (CMODE == WIDE_XOR) {
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if ((threadIdx.x & 63u) == 0 && xacc) atomicXor(reinterpret_cast<unsigned long long *>(a.out_a), (unsigned long long)xacc);
    }
}

// Minimizers (docs/src/replacements.md:33-51, test/benchmark.jl:96-110) of windows of `window_kmers` kmers, `stride` apart.
// The element's slot of out_a holds the current minimum while the window is walked, so no register array depends on the width.
// mode 0 (the published example, literally): each next symbol is shifted into the CURRENT MINIMUM; mode 1: the true
// sliding-window minimum of fx_hash over the window's kmers (first of equal hashes).
template <int SRC_BITS, int DST>
__global__ __launch_bounds__(BLOCK) void wide_minimizer_kernel(const StreamArgs a, const uint32_t n_words) {
    const uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n_kmers) return;
    const uint64_t start = g * a.stride;
    const uint64_t mask = head_mask((int)a.k, DST);
    uint64_t *best = a.out_a + g * n_words;
    uint64_t hash = 0;
    for (uint32_t w = 0; w < n_words; ++w) {
        best[w] = wide_word<SRC_BITS, DST>(a, start, n_words, w, false);
        hash = fx_step(hash, best[w]);
    }
    if (a.minimizer_mode == 0) {
        for (uint32_t off = 0; off + 1 < a.window_kmers; ++off) {
            const uint64_t sym = wide_symbol<SRC_BITS, DST>(a, start + a.k + off);
            // fx_hash of shift_encoding(best, sym) (construction_utils.jl:129-134), word by word
            uint64_t nh = 0;
            for (uint32_t w = 0; w < n_words; ++w) {
                uint64_t v = (best[w] << DST) | (w + 1 < n_words ? best[w + 1] >> (64 - DST) : sym);
                if (w == 0) v &= mask;
                nh = fx_step(nh, v);
            }
            if (nh < hash) {
                hash = nh;
                for (uint32_t w = 0; w < n_words; ++w) {  // in place, head first: word w + 1 is read before it is rewritten
                    uint64_t v = (best[w] << DST) | (w + 1 < n_words ? best[w + 1] >> (64 - DST) : sym);
                    if (w == 0) v &= mask;
                    best[w] = v;
                }
            }
        }
    } else {
        uint32_t best_off = 0;
        for (uint32_t off = 1; off < a.window_kmers; ++off) {
            uint64_t nh = 0;
            for (uint32_t w = 0; w < n_words; ++w) nh = fx_step(nh, wide_word<SRC_BITS, DST>(a, start + off, n_words, w, false));
            if (nh < hash) {
                hash = nh;
                best_off = off;
            }
        }
        if (best_off)
            for (uint32_t w = 0; w < n_words; ++w) best[w] = wide_word<SRC_BITS, DST>(a, start + best_off, n_words, w, false);
    }
}

// the six (source, kmer alphabet) pairs of RecodingScheme (construction.jl:75-100)
#define KMERS_WIDE_DISPATCH(CALL, src_bits, dst_bits) \
    do {                                              \
        if ((src_bits) == 8 && (dst_bits) == 2) CALL(8, 2);      \
        else if ((src_bits) == 8) CALL(8, 4);                     \
        else if ((src_bits) == 4 && (dst_bits) == 2) CALL(4, 2);  \
        else if ((src_bits) == 2 && (dst_bits) == 2) CALL(2, 2);  \
        else if ((src_bits) == 4 && (dst_bits) == 4) CALL(4, 4);  \
        else CALL(2, 4);                                          \
    } while (0)

}  // namespace kmers
