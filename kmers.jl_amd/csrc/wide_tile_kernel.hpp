// wide_tile_kernel.hpp -- kmers of more than four words (FwKmers, FwRvIterator, CanonicalKmers + fx_hash, SpacedKmers and the
// fused XOR / MinHash consumers over them): the tile form of wide_kernel.hpp.  Kmer{A,K,N} has no bound on N
// (src/kmer.jl:97-111); wide_kernel.hpp computes every word of every kmer from single symbols read from HBM (about 0.3 TB/s
// of output: profiles/r03_wide.md), this kernel stages a tile's symbols ONCE in LDS, recoded into the kmer alphabet
// (RecodingScheme, src/construction.jl:75-100: the stream kernel's own stage_word), and then treats the OUTPUT WORD as the
// work item:
//   * word w of the forward kmer of window g is 64 / DST consecutive symbols of the staged stream, reversed
//     (first symbol in the top bits: the Kmer layout, kmer.jl:32-51) -- two LDS reads, one funnel shift, one symbol reversal;
//     word w of the reverse complement is the complement of the mirrored stretch, unreversed (transformations.jl:1-34);
//   * consecutive lanes take consecutive words of the output array (kmer g's N words, then kmer g + 1's ...), so every wave
//     store is 512 contiguous bytes whatever N is, and no register array depends on N;
//   * canonical kmers: one lane per KMER first decides fw < rv (kmer.jl:176-178: the first differing word from the head,
//     almost always the head itself), folds fx_hash over the chosen strand's words (kmer.jl:255-260) and leaves the decision
//     in LDS for the word pass.
// SpacedKmers are the same with windows J symbols apart.  Minimizers and kmers of which not even one fits the LDS (K beyond
// ~240 000) stay on wide_kernel.hpp.
#pragma once
#include <algorithm>

#include "context.hpp"
#include "stream_kernel.hpp"

namespace kmers {

enum WideMode { WMODE_FW = 0, WMODE_CANON = 1, WMODE_XOR = 2, WMODE_SKETCH = 3 };

constexpr uint32_t WIDE_TILE_LDS_BYTES = 60u * 1024u;  // budget of the staged stream + the per-kmer decisions

// 64-bit words of LDS stream a tile of `tile` windows needs (the stream starts at the source word that holds the tile's
// first symbol; two spare words behind it: a chunk read may touch the word after its last symbol)
// (64-bit: a tile of a very long stride must compare as too large for the LDS budget, not wrap around to a small number)
inline uint64_t wide_tile_stream_words(uint32_t tile, uint32_t k, uint32_t stride, int src_bits, int dst_bits) {
    const uint64_t symbols = (uint64_t)(tile - 1u) * stride + k + (uint64_t)(64 / src_bits - 1);
    const uint64_t src_words = (symbols * (uint64_t)src_bits + 63u) / 64u + 1u;
    return (src_words * (uint64_t)dst_bits + (uint64_t)src_bits - 1u) / (uint64_t)src_bits + 2u;
}

template <int SRC_BITS, int DST, int WMODE>
__global__ __launch_bounds__(BLOCK) void wide_tile_kernel(const StreamArgs a, const uint32_t n_words, const uint32_t stream_words,
                                                           const uint32_t cw_pitch, const uint32_t vec16) {
    extern __shared__ uint64_t wl[];
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    uint8_t *const take = reinterpret_cast<uint8_t *>(wl + stream_words);  // per window of the tile: 1 = the forward strand is the canonical one
    // CANON with hashes, tiles whose words fit the LDS (cw_pitch != 0: words per window, odd): the word pass leaves the canonical
    // kmers' words here and the fx_hash fold reads them back, instead of the fold computing every word a second time
    uint64_t *const cw = reinterpret_cast<uint64_t *>(take + ((a.tile_kmers + 7u) & ~7u));
    constexpr uint32_t SPW = 64u / (uint32_t)DST;                          // symbols per kmer word
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k, T = a.tile_kmers, J = a.stride;  // window gl of a tile starts J * gl symbols behind the tile's first
    const uint32_t text = (SRC_BITS == 8 && DST == 2 && a.ascii_table <= 1u) ? 1u + a.ascii_table : 0u;
    if constexpr (SRC_BITS == 8) {
        if (!text) {
            for (uint32_t i = tid; i < 256u; i += BLOCK) lut[i] = ascii_entry(a.ascii_table, i);
            block_sync();
        }
    }
    uint64_t xacc = 0;  // WMODE_XOR: this lane's fold over all of its tiles
    // FW / CANON: one tile per workgroup (short-lived workgroups write fastest, profiles/r01_tuning.md); the consumers run a
    // persistent grid (launch_wide_tile) and walk the tiles
    // split order (a.split_order, one tile per workgroup): even workgroups walk the first half of the tiles, odd ones the second
    // -- two write windows half an array apart, for an output array that lies across a class boundary of HBM (stream_launch.hpp)
    const uint64_t n_first = (a.n_tiles + 1u) / 2u;
    for (uint64_t slot = blockIdx.x; slot < (a.split_order ? 2u * n_first : a.n_tiles); slot += gridDim.x) {
    const uint64_t tile_id = a.split_order ? ((slot & 1u) ? n_first + (slot >> 1) : (slot >> 1)) : slot;
    if (tile_id >= a.n_tiles) break;  // (the odd half is the shorter one: only the last slot can fall off, in a one-tile-per-workgroup grid)
    const uint64_t g0 = tile_id * T;
    const uint32_t nk = a.n_kmers - g0 < (uint64_t)T ? (uint32_t)(a.n_kmers - g0) : T;
    // ---- stage: every source word the tile's windows touch, recoded, at its own word-aligned place in the stream
    const uint64_t bit0 = a.first_bit + g0 * (uint64_t)J * (uint64_t)SRC_BITS;
    const uint64_t ws = bit0 >> 6;
    const uint32_t off0 = (uint32_t)(bit0 & 63u) / (uint32_t)SRC_BITS;  // symbols of the first staged word in front of the tile
    const uint32_t n_src = (uint32_t)(((bit0 + ((uint64_t)(nk - 1u) * J + k) * SRC_BITS + 63u) >> 6) - ws);
    for (uint32_t j = tid; j < n_src; j += BLOCK) {
        const uint64_t x = a.src[ws + j];
        const uint64_t f = stage_word<SRC_BITS, DST>(wl, j, x, lut, text);
        // (SpacedKmers with J >= K never inspects the symbols between two windows, SpacedKmers.jl:133-134: report_bad_symbols drops them)
        if (f) report_bad_symbols<SRC_BITS, false>(a.err_slot, a.first_bit, a.inspect_end, J, k, ws + j, f, x, a.err_origin);
    }
    block_sync();

    // c symbols of the tile from symbol `sym` on, little-endian (symbol j at bits DST * j)
    auto chunk = [&](uint32_t sym, uint32_t c) -> uint64_t {
        const uint32_t b = (off0 + sym) * (uint32_t)DST, q = b >> 6;
        const uint64_t v = funnel64(wl[q], wl[q + 1u], b & 63u);
        return c < SPW ? v & ((1ull << (c * (uint32_t)DST)) - 1ull) : v;
    };
    const uint32_t c_head = k - (n_words - 1u) * SPW;  // symbols in the head word (kmer.jl:128: the unused bits are its top bits)
    // word w (0 = head) of window gl's forward kmer, or of its reverse complement
    auto word = [&](uint32_t window, uint32_t w, bool rc) -> uint64_t {
        const uint32_t gl = window * J;  // the window's first symbol
        const uint32_t c = w == 0u ? c_head : SPW;
        const uint32_t p_lo = w == 0u ? 0u : k - (n_words - w) * SPW;  // the window's symbol in the word's top position
        if (!rc) return rev_symbols<DST>(chunk(gl + p_lo, c)) >> (64u - c * (uint32_t)DST);
        const uint64_t v = comp_symbols<DST>(chunk(gl + k - p_lo - c, c));
        return c < SPW ? v & ((1ull << (c * (uint32_t)DST)) - 1ull) : v;
    };

    if constexpr (WMODE != WMODE_FW) {
        // ---- one lane per kmer: the canonical strand, its hash, the consumers
        const bool canonical = WMODE != WMODE_XOR || a.xor_canonical != 0;
        const bool want_hash = WMODE == WMODE_SKETCH || (WMODE == WMODE_CANON && (a.out_b || a.tuples) && cw_pitch == 0u);
        uint64_t threshold = 0;
        if constexpr (WMODE == WMODE_SKETCH)
            threshold = a.threshold_ptr ? __hip_atomic_load(a.threshold_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.threshold;
        for (uint32_t gl = tid; gl < nk; gl += BLOCK) {
            bool take_fw = !canonical;  // fw < rv ? fw : rv (CanonicalKmers.jl:224): equal strands yield rv, the same words
            if (canonical) {
                for (uint32_t w = 0; w < n_words; ++w) {
                    const uint64_t f = word(gl, w, false), r = word(gl, w, true);
                    if (f != r) {
                        take_fw = f < r;
                        break;
                    }
                }
            }
            if constexpr (WMODE == WMODE_XOR) {
                xacc ^= word(gl, 0u, !take_fw);  // the reducer of test/benchmark.jl:9-15: data[1]
                continue;
            }
            if constexpr (WMODE == WMODE_CANON) take[gl] = take_fw ? 1u : 0u;
            if (want_hash) {
                uint64_t h = a.seed;
                for (uint32_t w = 0; w < n_words; ++w) h = fx_step(h, word(gl, w, !take_fw));
                if constexpr (WMODE == WMODE_SKETCH) {
                    if (h < threshold) sketch_candidate(a, h);
                } else if (a.tuples) {
                    a.out_a[(g0 + gl) * (n_words + 1u) + n_words] = h;  // Tuple{Kmer,UInt64}: the hash behind the kmer's words
                } else {
                    a.out_b[g0 + gl] = h;
                }
            }
        }
        block_sync();  // the decisions are in LDS / every lane is done with the staged stream
        if constexpr (WMODE != WMODE_CANON) continue;
        if (!a.out_a && cw_pitch == 0u) continue;
    }

    // ---- one lane per OUTPUT WORD: consecutive lanes write consecutive words of the array
    // element = the words one window contributes to out_a: N (separate arrays; the hash / the reverse complement go to out_b),
    // 2N (Tuple{Kmer,Kmer}: forward words, then reverse complement), N of N + 1 (Tuple{Kmer,UInt64}: the hash was stored above)
    const uint32_t per = (WMODE == WMODE_FW && a.tuples) ? 2u * n_words : n_words;
    const uint32_t pitch = (WMODE == WMODE_CANON && a.tuples) ? n_words + 1u : per;  // words between two elements in out_a
    // Where the tile's words are one contiguous, 16-byte aligned stretch of the array (separate arrays / Tuple{Kmer,Kmer}, an even
    // first word, aligned bases: vec16) a lane takes TWO consecutive words and stores them at once -- the second may be the first
    // word of the next kmer -- like the 16-byte stores of the compile-time-width kernels.
    const uint64_t first_word = g0 * per;
    if (vec16 && pitch == per && (first_word & 1u) == 0u) {
        const uint32_t total = nk * per;
        const uint32_t step2_g = (2u * (uint32_t)BLOCK) / per, step2_w = (2u * (uint32_t)BLOCK) % per;
        uint32_t g2 = (2u * tid) / per, w2 = (2u * tid) % per;
        for (uint32_t i = 2u * tid; i < total; i += 2u * (uint32_t)BLOCK) {
            uint32_t g3 = g2, w3 = w2 + 1u;  // the second word of the pair
            if (w3 == per) {
                w3 = 0;
                ++g3;
            }
            const bool two = i + 1u < total;
            uint64_t va = 0, vb = 0, ra = 0, rb = 0;
            if constexpr (WMODE == WMODE_FW) {
                if (a.tuples) {
                    va = w2 < n_words ? word(g2, w2, false) : word(g2, w2 - n_words, true);
                    if (two) vb = w3 < n_words ? word(g3, w3, false) : word(g3, w3 - n_words, true);
                } else {
                    if (a.out_a) {
                        va = word(g2, w2, false);
                        if (two) vb = word(g3, w3, false);
                    }
                    if (a.out_b) {
                        ra = word(g2, w2, true);
                        if (two) rb = word(g3, w3, true);
                    }
                }
            } else {
                va = word(g2, w2, take[g2] == 0u);
                if (two) vb = word(g3, w3, take[g3] == 0u);
                if (cw_pitch) {
                    cw[g2 * cw_pitch + w2] = va;
                    if (two) cw[g3 * cw_pitch + w3] = vb;
                }
            }
            const uint64_t at = first_word + i;
            if (two) {
                if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + at) = make_ulonglong2(va, vb);
                if (WMODE == WMODE_FW && !a.tuples && a.out_b) *reinterpret_cast<ulonglong2 *>(a.out_b + at) = make_ulonglong2(ra, rb);
            } else {
                if (a.out_a) a.out_a[at] = va;
                if (WMODE == WMODE_FW && !a.tuples && a.out_b) a.out_b[at] = ra;
            }
            g2 += step2_g;
            w2 += step2_w;
            if (w2 >= per) {
                w2 -= per;
                ++g2;
            }
        }
    } else {
    const uint32_t step_g = (uint32_t)BLOCK / per, step_w = (uint32_t)BLOCK % per;
    uint32_t gl = tid / per, w = tid % per;
    while (gl < nk) {
        const uint64_t at = (g0 + gl) * pitch + w;
        if constexpr (WMODE == WMODE_FW) {
            if (a.tuples) {
                a.out_a[at] = w < n_words ? word(gl, w, false) : word(gl, w - n_words, true);
            } else {
                if (a.out_a) a.out_a[at] = word(gl, w, false);
                if (a.out_b) a.out_b[at] = word(gl, w, true);
            }
        } else {
            const uint64_t v = word(gl, w, take[gl] == 0u);
            if (a.out_a) a.out_a[at] = v;
            if (cw_pitch) cw[gl * cw_pitch + w] = v;
        }
        gl += step_g;
        w += step_w;
        if (w >= per) {
            w -= per;
            ++gl;
        }
    }
    }
    if constexpr (WMODE == WMODE_CANON) {
        if (cw_pitch) {
            // ---- fx_hash(canonical kmer, seed): the fold over the words the word pass left in LDS (kmer.jl:255-260), one lane per kmer
            block_sync();
            for (uint32_t g = tid; g < nk; g += BLOCK) {
                uint64_t h = a.seed;
                for (uint32_t j = 0; j < n_words; ++j) h = fx_step(h, cw[g * cw_pitch + j]);
                if (a.tuples) a.out_a[(g0 + g) * (n_words + 1u) + n_words] = h;
                else a.out_b[g0 + g] = h;
            }
        }
    }
    if (slot + gridDim.x < (a.split_order ? 2u * n_first : a.n_tiles)) block_sync();  // the next tile restages the stream
    }
    if constexpr (WMODE == WMODE_XOR) {
        // wavefront XOR-reduce (64 lanes), then one atomic per wave and workgroup
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if ((tid & 63u) == 0 && xacc) atomicXor(reinterpret_cast<unsigned long long *>(a.out_a), (unsigned long long)xacc);
    }
}

// Launch of the tile form; returns -1 (nothing launched) where it does not apply -- a kmer so long that not even one window
// fits the LDS budget -- and the caller goes on to wide_kernel.hpp.  Strided iteration (SpacedKmers) stages the symbols between
// its windows too: with J far above K the tiles shrink to the windows that fit (down to one), which is still one coalesced read
// of the stretch instead of K single-symbol reads per window.
template <int WMODE>
int launch_wide_tile(kmers_ctx *ctx, StreamArgs &a, int src_bits, int dst_bits, uint32_t n_words) {
    if (a.n_kmers == 0) return -1;
    // FW / CANON stream N words per window: 16 KiB per output array and workgroup, like the stream kernels; the consumers
    // store nothing: long tiles
    const bool streams = WMODE == WMODE_FW || WMODE == WMODE_CANON;
    // (CANON first walks its tile one lane per KMER -- the strand decision and the fx_hash fold: whole multiples of the
    // workgroup, however few windows 16 KiB are)
    uint32_t tile = WMODE == WMODE_FW ? std::max<uint32_t>(1u, (2048u / n_words) & ~1u)  // (even: a tile's first output word is)
                  : WMODE == WMODE_CANON ? std::max<uint32_t>((uint32_t)BLOCK, (2048u / n_words + BLOCK - 1u) / BLOCK * BLOCK) : 2048u;
    if (ctx->tile_kmers > 0) tile = (uint32_t)std::min<int64_t>(ctx->tile_kmers, 1 << 16);  // tests, tuning
    tile = (uint32_t)std::min<uint64_t>(tile, a.n_kmers);
    auto lds_bytes = [&](uint32_t t) { return wide_tile_stream_words(t, a.k, a.stride, src_bits, dst_bits) * 8u + ((t + 7u) & ~7u); };
    const uint32_t word_pitch = n_words | 1u;  // (odd: the fold's lanes read LDS words `pitch` apart)
    while (tile > 1u && lds_bytes(tile) > WIDE_TILE_LDS_BYTES) tile /= 2u;
    if (lds_bytes(tile) > WIDE_TILE_LDS_BYTES) return -1;
    const uint64_t n_tiles = (a.n_kmers + tile - 1u) / tile;
    if (n_tiles >= (1ull << 31)) return -1;
    a.tile_kmers = tile;
    a.n_tiles = n_tiles;
    const uint32_t sw = (uint32_t)wide_tile_stream_words(tile, a.k, a.stride, src_bits, dst_bits);  // (fits: lds_bytes(tile) is within the budget)
    size_t dyn = lds_bytes(tile);
    uint32_t cw_pitch = 0;  // CANON with hashes: the tile's canonical words staged in LDS for the fold, if they fit
    if (WMODE == WMODE_CANON && (a.out_b || a.tuples) && dyn + (size_t)tile * word_pitch * 8u <= WIDE_TILE_LDS_BYTES) {
        cw_pitch = word_pitch;
        dyn += (size_t)tile * word_pitch * 8u;
    }
    // ONE output array (FwKmers, SpacedKmers; CanonicalKmers, whose hashes are a small second stream) whose halves lie in two region
    // classes (a block of the pool taken by role, kmers_dev_alloc_role): two write windows, like the stream kernels' lone outputs
    const size_t element_bytes = (size_t)8 * (a.tuples ? (WMODE == WMODE_FW ? 2u * n_words : n_words + 1u) : n_words);
    const bool lone = streams && ctx->max_grid <= 0 && ctx->split_order >= 0 && a.out_a && (WMODE == WMODE_CANON || !a.out_b) &&
                      pool_halves_differ(ctx, a.out_a, (size_t)a.n_kmers * element_bytes) >= 0.75f;
    a.split_order = (lone || (streams && ctx->max_grid <= 0 && ctx->split_order > 0)) && n_tiles >= 2 ? 1u : 0u;
    const uint64_t slots = a.split_order ? 2u * ((n_tiles + 1u) / 2u) : n_tiles;
    const uint64_t resident = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (streams ? slots : (uint64_t)ctx->n_cus * 8u);
    dim3 grid((unsigned)std::min<uint64_t>(slots, resident)), block(BLOCK);
    // 16-byte stores where the output bases allow them (the kernel checks the tile's first word)
    const uint32_t vec16 = ((!a.out_a || (reinterpret_cast<uintptr_t>(a.out_a) & 15u) == 0) &&
                            (a.tuples || WMODE != WMODE_FW || !a.out_b || (reinterpret_cast<uintptr_t>(a.out_b) & 15u) == 0)) ? 1u : 0u;
#define WIDET(SB, DB)                                                                                                              \
    do {                                                                                                                           \
        if (dyn > 48u * 1024u)                                                                                                     \
            HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(wide_tile_kernel<SB, DB, WMODE>),                      \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)WIDE_TILE_LDS_BYTES));               \
        hipLaunchKernelGGL((wide_tile_kernel<SB, DB, WMODE>), grid, block, dyn, ctx->stream, a, n_words, sw, cw_pitch, vec16);    \
    } while (0)
    if (src_bits == 8 && dst_bits == 2) WIDET(8, 2);
    else if (src_bits == 8) WIDET(8, 4);
    else if (src_bits == 4 && dst_bits == 2) WIDET(4, 2);
    else if (src_bits == 2 && dst_bits == 2) WIDET(2, 2);
    else if (src_bits == 4 && dst_bits == 4) WIDET(4, 4);
    else WIDET(2, 4);
#undef WIDET
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

}  // namespace kmers
