// composition_kernel.hpp -- the k-mer composition recipe of docs/src/composition.md:28-39,
//     for kmer in FwDNAMers{K}(seq); counts[as_integer(kmer) + 1] += 1; end
// as one fused pass: nothing is materialised per kmer, the counters live in LDS.
//
// Shape of the problem on MI355X: 1 Gbase is only 0.25-1 GB of source (HBM: ~0.1 ms) but 1e9
// scattered increments.  Global atomics execute at the memory side (about 27 G/s for scattered
// dwords, measured: 36.6 ms per Gbase), LDS atomics at about one wave-instruction per few clocks
// per CU.  So every workgroup keeps a private histogram in LDS and adds it to the global counters
// when it retires:
//   * 1024-thread workgroups (16 wavefronts hide the LDS-atomic and source-load latency even when
//     the histogram leaves room for only one workgroup per CU);
//   * 16-bit counters packed two per LDS word: 4^8 = 65 536 bins fit in 128 KiB.  A counter can
//     only overflow after 65 535 increments, so the workgroup tracks an upper bound on its largest
//     counter (bound += kmers of the tile); when the bound gets close it measures the true maximum
//     (one sweep over the histogram) and flushes to the global counters only if a real counter is
//     close to the limit -- never on uniform data, often (but cheaply: few non-zero bins) on
//     low-complexity data;
//   * K > 8: the bins are split into passes of 65 536; pass p counts the kmers whose index has
//     p in its upper bits (the host launches 4^(K-8) passes);
//   * phase 2 takes 16 consecutive kmers from ONE 64-bit window of the 2-bit stream: the symbol-
//     reversed window R holds kmer j of the lane at bits [64-2K-2j, 64-2j), so a kmer costs a
//     shift, a mask and the LDS atomic.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

constexpr int CBLOCK = 1024;                       // threads per workgroup
constexpr int CRUN = 16;                           // consecutive kmers per lane (one 64-bit window: K <= 16)
constexpr int CTILE = CBLOCK * CRUN;               // kmers per tile = 16384 symbols of 2-bit stream (4 KiB)
constexpr uint32_t CBINS_LOG2 = 16;                // bins per pass
constexpr uint32_t CHIST_WORDS = 1u << (CBINS_LOG2 - 1);  // two 16-bit counters per word
constexpr uint32_t COUNTER_LIMIT = 0xFFFFu;

struct CompositionArgs {
    const uint64_t *src;
    uint64_t first_bit;
    uint64_t n_bases;
    uint64_t n_kmers;
    uint64_t n_tiles;
    uint32_t *counts;              // 4^K global counters (zeroed by the host)
    unsigned long long *err_slot;
    uint64_t err_origin;           // added to reported positions (kmers_seq.index_origin)
    uint32_t ascii_table;
    uint32_t k;
    uint32_t pass;                 // upper bits of the indices this launch counts (0 when 4^K <= 65536)
    uint32_t hist_words;           // LDS words in use: max(1, min(4^K, 65536) / 2)
};

// bin b of the pass lives in half (b >> half_shift) of word (b & (hist_words - 1)): both halves of
// the histogram are contiguous runs of bins, so a flush is two fully coalesced atomic instructions.
template <int SRC_BITS>
__global__ __launch_bounds__(CBLOCK) void composition_kernel(const CompositionArgs a) {
    extern __shared__ uint32_t hist[];                      // a.hist_words words
    __shared__ uint64_t lds[CTILE * 2 / 64 + 16];           // the tile's 2-bit stream
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    __shared__ uint32_t wave_max[CBLOCK / 64];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint32_t words = a.hist_words;
    const uint32_t half_shift = 31u - (uint32_t)__builtin_clz(words);  // log2(words); bins per pass = 2 * words
    const uint32_t bins_log2 = 2u * k < CBINS_LOG2 ? 2u * k : CBINS_LOG2;
    const uint64_t kmask = (k >= 32) ? ~0ull : ((1ull << (2u * k)) - 1ull);
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += CBLOCK) lut[i] = ascii_entry(a.ascii_table, i);
    }
    for (uint32_t i = tid; i < words; i += CBLOCK) hist[i] = 0;
    uint32_t bound = 0;  // upper bound on the largest 16-bit counter of this workgroup (uniform)

    // add the histogram to the global counters and clear it
    auto flush = [&]() {
        uint32_t *out = a.counts + ((size_t)a.pass << CBINS_LOG2);
        for (uint32_t i = tid; i < words; i += CBLOCK) {
            const uint32_t v = hist[i];
            if (v & 0xFFFFu) atomicAdd(out + i, v & 0xFFFFu);
            if (v >> 16) atomicAdd(out + i + words, v >> 16);
            hist[i] = 0;
        }
    };

    // geometry of a tile: first source word, symbol offset inside it, source words, kmers
    struct Geo { uint64_t w0; uint32_t b0, nw, mt; };
    auto geometry = [&](uint64_t tile) {
        Geo g;
        const uint64_t m0 = tile * CTILE;
        const uint64_t left = a.n_kmers - m0;
        g.mt = left < (uint64_t)CTILE ? (uint32_t)left : (uint32_t)CTILE;
        const uint64_t bit0 = a.first_bit + m0 * SRC_BITS;
        g.w0 = bit0 >> 6;
        g.b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(g.mt - 1) + k) * SRC_BITS;
        g.nw = (uint32_t)(((end_bit + 63) >> 6) - g.w0);
        return g;
    };
    // source words of a tile per thread: 16384 + K - 1 + 31 symbols over 1024 threads
    constexpr int PRE = (CTILE + 64) * SRC_BITS / 64 / CBLOCK + 1;
    uint64_t pre[PRE];
    auto prefetch = [&](const Geo &g) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t wi = tid + (uint32_t)i * CBLOCK;
            pre[i] = wi < g.nw ? a.src[g.w0 + wi] : 0;
        }
    };

    uint64_t tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(geometry(tile));
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const Geo g = geometry(tile);
        const uint32_t mt = g.mt, b0 = g.b0;

        if (bound + mt > COUNTER_LIMIT) {
            // a counter MIGHT overflow during this tile: measure the true maximum
            block_sync();  // the previous tile's increments must have landed (see device_bits.hpp)
            uint32_t mx = 0;
            for (uint32_t i = tid; i < words; i += CBLOCK) {
                const uint32_t v = hist[i];
                mx = max(mx, max(v & 0xFFFFu, v >> 16));
            }
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off, 64));
            if ((tid & 63u) == 0) wave_max[tid >> 6] = mx;
            block_sync();
            mx = 0;
#pragma unroll
            for (int w = 0; w < CBLOCK / 64; ++w) mx = max(mx, wave_max[w]);
            bound = mx;
            if (bound + (uint32_t)CTILE > COUNTER_LIMIT) {  // a real counter is close: flush
                flush();
                bound = 0;
            }
        }
        bound += mt;

        block_sync();  // previous tile's readers are done with the stream; histogram zeroing / flush visible
        // ---- phase 1: source words (already in registers) -> 2-bit stream in LDS (RecodingScheme,
        //      construction.jl:75-100); then the NEXT tile's words start their trip from HBM
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t wi = tid + (uint32_t)i * CBLOCK;
            if (wi < g.nw) {
                const uint64_t f = stage_word<SRC_BITS, 2>(lds, wi, pre[i], lut);
                if constexpr (SRC_BITS != 2) {
                    if (f) report_bad_symbols<SRC_BITS, true>(a.err_slot, a.first_bit, a.n_bases, 1u, k, g.w0 + wi, f, pre[i], a.err_origin);
                }
            }
        }
        if (tile + gridDim.x < a.n_tiles) prefetch(geometry(tile + gridDim.x));
        block_sync();

        // ---- phase 2: 16 consecutive forward kmers per lane from one 64-bit window ---------
        const uint32_t r0 = tid * CRUN;
        if (r0 < mt) {
            const uint32_t bit = 2u * (r0 + b0);
            const uint32_t q = bit >> 6, s = bit & 63u;
            // symbols r0 .. r0+31, little-endian; bits past the staged stream only reach kmers j >= cnt
            const uint64_t W = funnel64(lds[q], lds[q + 1], s);
            // symbol r0 in the top two bits; shifted so that kmer j of the lane = bits [2(15-j), 2(15-j)+2K):
            // as_integer(kmer) (kmer.jl:305-326), first symbol most significant.  K <= 16: 32-bit fields.
            const uint64_t R = rev2(W) >> (64u - 2u * k - 2u * (CRUN - 1));
            const uint32_t lo = (uint32_t)R, hi = (uint32_t)(R >> 32);
            const uint32_t cnt = mt - r0 < (uint32_t)CRUN ? mt - r0 : (uint32_t)CRUN;
            const uint32_t kmask32 = (uint32_t)kmask, bin_mask = (1u << bins_log2) - 1u, word_mask = words - 1u;
            auto count_one = [&](uint32_t idx) {
                const uint32_t b = idx & bin_mask;
                const uint32_t inc = ((b >> half_shift) & 1u) * 0xFFFFu + 1u;  // 1 or 1 << 16
                __hip_atomic_fetch_add(&hist[b & word_mask], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            };
            const bool multipass = 2u * k > CBINS_LOG2;
            if (cnt == (uint32_t)CRUN && !multipass) {  // the common case: no per-kmer predicate
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)CRUN; ++j) {
                    const uint32_t off = 2u * (CRUN - 1u - j);
                    count_one((off ? __builtin_amdgcn_alignbit(hi, lo, off) : lo) & kmask32);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < (uint32_t)CRUN; ++j) {
                    const uint32_t off = 2u * (CRUN - 1u - j);
                    const uint32_t idx = (off ? __builtin_amdgcn_alignbit(hi, lo, off) : lo) & kmask32;
                    if (j < cnt && (idx >> bins_log2) == a.pass) count_one(idx);
                }
            }
        }
    }
    block_sync();
    flush();
}

}  // namespace kmers
