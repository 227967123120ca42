// stream_launch.hpp -- host-side launcher of stream_kernel (tile choice, template dispatch).  A template over the kernel's MODE:
// each translation unit instantiates only the modes it launches (iterators_api.hip: FW, CANON; consumers_api.hip: XOR, SKETCH,
// COUNT, MINIMIZER), so no kernel is compiled twice.
#pragma once
#include "api_common.hpp"

namespace kmers {

// Default tile: about 16 KiB of output per workgroup (four 16-byte stores per lane), one tile
// per workgroup.  Measured on MI355X (profiles/r01_tuning.md): shorter workgroups are bound by
// workgroup launch + the exposed source-load latency, longer ones and persistent grid-stride
// loops lose 10-20 % of the HBM write rate.
// Measured in round 3 (profiles/r03_tuning.md): more than one tile per visit LOSES -- SpacedDNAMers{21,3}, 1 Gbase: 0.757 / 0.746 /
// 0.733 / 0.728 / 0.698 of 8 TB/s at 1 / 2 / 3 / 4 / 6 tiles of 2048 per visit -- a workgroup that lives longer writes slower, and
// that costs more than the load round it hides.  The mechanism stays (KMERS_PARAM_SUBTILES) for sources with slower loads.
constexpr int64_t DEFAULT_SUBTILES = 1;

inline uint32_t default_tile(uint32_t out_bytes_per_kmer, uint32_t pass) {
    uint32_t t = (16384u / std::max<uint32_t>(out_bytes_per_kmer, 1u)) / pass * pass;
    return std::max<uint32_t>(pass, t);
}

template <int MODE, int SB, int DB>
void launch_widths(int n_words, bool s1, bool pair, bool fwd, dim3 grid, dim3 block, hipStream_t st, const StreamArgs &a, size_t dyn_lds) {
    if constexpr (MODE == MODE_FW && DB == 2) {
        if (fwd) {  // forward kmers only: kmer-order staging (stream_kernel.hpp, FWD)
            if (pair) {
                hipLaunchKernelGGL((stream_kernel<SB, DB, 1, MODE, false, false, true, true>), grid, block, dyn_lds, st, a);
                return;
            }
#define LAUNCH_FWD(NN)                                                                                                    \
    do {                                                                                                                  \
        if (s1) hipLaunchKernelGGL((stream_kernel<SB, DB, NN, MODE, true, false, false, true>), grid, block, dyn_lds, st, a);  \
        else hipLaunchKernelGGL((stream_kernel<SB, DB, NN, MODE, false, false, false, true>), grid, block, dyn_lds, st, a);    \
    } while (0)
            switch (n_words) {
                case 1: LAUNCH_FWD(1); break;
                case 2: LAUNCH_FWD(2); break;
                case 3: LAUNCH_FWD(3); break;
                default: LAUNCH_FWD(4); break;
            }
#undef LAUNCH_FWD
            return;
        }
    }
    if constexpr (MODE == MODE_FW || MODE == MODE_CANON) {
        if (a.tuples) {  // array-of-structs outputs: one kmer per lane per pass
            switch (n_words) {
                case 1: hipLaunchKernelGGL((stream_kernel<SB, DB, 1, MODE, false, true>), grid, block, dyn_lds, st, a); break;
                case 2: hipLaunchKernelGGL((stream_kernel<SB, DB, 2, MODE, false, true>), grid, block, dyn_lds, st, a); break;
                case 3: hipLaunchKernelGGL((stream_kernel<SB, DB, 3, MODE, false, true>), grid, block, dyn_lds, st, a); break;
                default: hipLaunchKernelGGL((stream_kernel<SB, DB, 4, MODE, false, true>), grid, block, dyn_lds, st, a); break;
            }
            return;
        }
    }
    if constexpr (MODE == MODE_FW || MODE == MODE_XOR) {
        if (pair) {  // strided one-word kmers, two lattice kmers per lane (16-byte stores)
            hipLaunchKernelGGL((stream_kernel<SB, DB, 1, MODE, false, false, true>), grid, block, dyn_lds, st, a);
            return;
        }
    }
#define LAUNCH(NN)                                                                                   \
    do {                                                                                             \
        if (s1) hipLaunchKernelGGL((stream_kernel<SB, DB, NN, MODE, true>), grid, block, dyn_lds, st, a);  \
        else hipLaunchKernelGGL((stream_kernel<SB, DB, NN, MODE, false>), grid, block, dyn_lds, st, a);    \
    } while (0)
    switch (n_words) {
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        case 3: LAUNCH(3); break;
        default: LAUNCH(4); break;
    }
#undef LAUNCH
}

template <int MODE>
int launch_stream(kmers_ctx *ctx, StreamArgs &a, int src_bits, int dst_bits, int n_words, bool vec_ok, size_t dyn_lds = 0) {
    // the tile kernel is instantiated for one to four words; wider kmers belong to wide_kernel.hpp / the run-time-width
    // single-pass kernel and a caller that comes here with them must hear about it
    if (n_words < 1 || n_words > 4) return fail(ctx, KMERS_E_UNSUPPORTED, "internal: the tile kernel takes kmers of one to four words");
    const uint32_t J = a.stride;
    const bool stride1 = (J == 1) && vec_ok && !a.tuples;
    const bool pair = (MODE == MODE_FW || MODE == MODE_XOR) && J > 1 && vec_ok && !a.tuples && n_words == 1;
#ifdef KMERS_NO_FWD  // A/B builds only
    const bool fwd = false;
#else
    const bool fwd = MODE == MODE_FW && dst_bits == 2 && !a.out_b && !a.tuples;  // no reverse complements wanted
#endif
    // ---- launch shape: threads per workgroup and kmers per tile (one tile per workgroup) -----------------------------------
    // The base rule (round 1): about 16 KiB of output per workgroup, 256 threads.  Round 3 measured every shape below against
    // it with the outputs in KNOWN places (profiles/r03_tuning.md sections 2, 5, 6, 7; fractions of 8 TB/s, fresh processes):
    //
    //   launch                                            | placement the launcher sees          | threads x kmers      | gain
    //   one-word elements, two arrays (C2 headline, FwRv, | well placed (pool_arrays_differ)     | 128 x 1536 (24 KiB)  | 0.82-0.84 -> 0.88-0.90
    //     kmers + starts)                                 | anything else                        | 256 x 1024 (rule)    | (24 KiB bimodal there)
    //   two-word kmers + reverse complements (C4)         | well placed                          | 256 x 768            | 0.87 -> 0.90-0.915
    //   three- / four-word kmers + reverse complements    | well placed                          | 256 x 512            | 0.75 -> 0.80 (four-word)
    //   two-word canonical kmers + hashes                 | well placed                          | 128 x 768            | 0.71 -> 0.87-0.88
    //                                                     | anything else                        | 128 x 512            | 0.70 -> 0.80-0.85
    //   three- / four-word canonical kmers + hashes       | any                                  | 128 x 512            | 0.58 -> 0.69 (four-word)
    //   ONE output array whose halves lie in two classes  | pool_halves_differ                   | 128 x 16 KiB, split  | C3 0.81 -> 0.87, two-word 0.80 -> 0.86
    //     (kmers_dev_alloc_role LONE_OUTPUT)              |   ... strided (SpacedKmers, C5)      | 256 x 40 KiB, split  | 0.75 -> 0.78-0.79
    //                                                     |   ... tuple elements                 | 256 x 6 x rule, split| 0.61 -> 0.82-0.83
    //   everything else                                   |                                      | 256 x rule           |
    //
    // "well placed" = both arrays in blocks of the device's class pool (pool_api.hip) that differ in region class at 90 % of 64
    // relative positions -- true by construction for arrays allocated one after the other; "split" = two write windows half an
    // array apart (stream_kernel.hpp, SPLIT ORDER).  Plain allocations get the base rule.  KMERS_PARAM_TILE_KMERS /
    // _BLOCK_THREADS / _SPLIT_ORDER override the table (tests, tools/).  (Rounds 3-5 also asked the map of a reserved arena and
    // timed the table against the rule once per placement: gone with the arena, round 6.)
    const bool materialises = MODE == MODE_FW || MODE == MODE_CANON;
    const bool two_arrays = materialises && stride1 && a.out_a && !a.tuples;
    const bool one_word_pair = two_arrays && n_words == 1 && (a.out_b || a.out_starts);
    const void *const second = a.out_b ? (const void *)a.out_b : (const void *)a.out_starts;
    const size_t bytes_a = (size_t)a.n_kmers * 8u * (size_t)n_words;
    const size_t bytes_b = (size_t)a.n_kmers * (MODE == MODE_CANON || !a.out_b ? 8u : 8u * (size_t)n_words);
    constexpr float PLACED = 0.9f;
    const bool pool_pair = materialises && a.out_a && second && pool_arrays_differ(ctx, a.out_a, bytes_a, second, bytes_b) >= PLACED;
    const bool spread = one_word_pair && pool_pair;
    const bool fwrc_wide = two_arrays && MODE == MODE_FW && n_words >= 2 && a.out_b && pool_pair;
    const bool canon_wide = two_arrays && MODE == MODE_CANON && n_words >= 2 && a.out_b;
    const bool canon2_spread = canon_wide && n_words == 2 && pool_pair;
    const uint32_t lone_bytes = a.tuples ? (MODE == MODE_FW ? 16u * n_words : 8u * n_words + 8u) : 8u * n_words;
    // ONE output array whose two halves lie in different classes (a block of the pool made that way by role) is written through
    // two windows half an array apart
    const bool lone_output = materialises && a.out_a && !a.out_b && !a.out_starts && ctx->split_order >= 0;
    const bool lone = lone_output && pool_halves_differ(ctx, a.out_a, (size_t)a.n_kmers * lone_bytes) >= 0.75f;  // (whole handles: short arrays cannot do better)
    uint32_t threads = ctx->block_threads > 0 ? (uint32_t)ctx->block_threads
                                              : ((spread || canon_wide || (lone && !a.tuples && J == 1)) ? 128u : (uint32_t)BLOCK);
    if (threads != 64u && threads != 128u) threads = (uint32_t)BLOCK;
    const uint32_t pass = ((stride1 || pair) && n_words == 1 ? 2u : 1u) * threads;  // kmers per workgroup pass
    uint32_t out_bytes = 8u * n_words * ((a.out_a ? 1u : 0u) + (MODE == MODE_FW && a.out_b ? 1u : 0u)) +
                         (MODE == MODE_CANON && a.out_b ? 8u : 0u) + (MODE == MODE_FW && a.out_starts ? 8u : 0u);
    if (a.tuples) out_bytes = MODE == MODE_FW ? 16u * n_words : 8u * n_words + 8u;
    if (MODE == MODE_XOR || MODE == MODE_SKETCH || MODE == MODE_COUNT) out_bytes = 4u;  // nothing streamed out: long tiles
    if (MODE == MODE_MINIMIZER) out_bytes = 8u * n_words;
    uint32_t max_tile_symbols = (uint32_t)MAX_TILE_BITS / (uint32_t)dst_bits;
    if (MODE == MODE_MINIMIZER) max_tile_symbols -= std::min<uint32_t>(max_tile_symbols / 2, a.window_kmers);  // room for the longer overlap
    uint32_t tile = ctx->tile_kmers > 0 ? (uint32_t)ctx->tile_kmers : default_tile(out_bytes, pass);
    if (ctx->tile_kmers <= 0) {  // the table above
        if (spread) tile = tile * 3u / 2u / pass * pass;
        else if (fwrc_wide) tile = n_words == 2 ? tile * 3u / 2u / pass * pass : 512u;
        else if (canon_wide) tile = canon2_spread ? 768u : 512u;
        else if (lone && a.tuples) tile *= 6u;
        else if (lone && J > 1) tile = tile * 5u / 2u;  // (clamped to what the LDS stream holds below: 5120 kmers at J = 3)
    }
    // (strided launches in ONE class: round 2's 32 KiB tile lost to 16 KiB on every box of round 3, 0.70-0.74 against 0.73-0.76)
    const uint32_t pass_now = pass;
    tile = std::min<uint32_t>(tile, max_tile_symbols / J);
    tile = std::max<uint32_t>(pass_now, tile / pass_now * pass_now);
    if ((uint64_t)(tile - 1) * J + 1 > (uint64_t)max_tile_symbols) return fail(ctx, KMERS_E_UNSUPPORTED, "stride too large for the tile kernel");
    a.tile_kmers = tile;
    a.n_tiles = (a.n_kmers + tile - 1) / tile;
    // strided kernels: consecutive tiles per workgroup visit, the next tile's source words in flight behind the current tile's
    // stores (stream_kernel.hpp; profiles/r03_tuning.md)
    a.subtiles = J > 1 ? (uint32_t)(ctx->subtiles > 0 ? ctx->subtiles : DEFAULT_SUBTILES) : 1u;
    a.stamps = reinterpret_cast<uint64_t *>(ctx->stamps_ptr);
    uint64_t cap = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (uint64_t)1 << 30;
    const uint64_t visits = (a.n_tiles + a.subtiles - 1) / a.subtiles;
    // two write windows per output array (stream_kernel.hpp, SPLIT ORDER): for a lone output across a class boundary, or when
    // KMERS_PARAM_SPLIT_ORDER asks for it everywhere
    a.split_order = visits >= 2 && ((materialises || MODE == MODE_MINIMIZER) && ctx->split_order > 0 || lone) ? 1u : 0u;
    const uint64_t slots = a.split_order ? 2 * ((visits + 1) / 2) : visits;
    dim3 grid((unsigned)std::min<uint64_t>(slots, cap));
    dim3 block(threads);
    ctx->last_threads = (int)threads;
    ctx->last_tile = (int)tile;
    ctx->last_split = (int)a.split_order;
    if (src_bits == 8 && dst_bits == 2) launch_widths<MODE, 8, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 8) launch_widths<MODE, 8, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 4 && dst_bits == 2) launch_widths<MODE, 4, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 2 && dst_bits == 2) launch_widths<MODE, 2, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 4 && dst_bits == 4) launch_widths<MODE, 4, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else launch_widths<MODE, 2, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

}  // namespace kmers
