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
    // Launch shape of the two-output, one-word-element launches (canonical kmers + hashes -- the headline --, forward + reverse
    // complements, forward kmers + start indices), measured in round 3 (profiles/r03_tuning.md, 1 Gbase, fresh processes):
    //   * the two arrays in well-separated region classes of HBM -- which the launcher knows for blocks of the context's arena
    //     (kmers_arena_spread) --: 24 KiB of output per workgroup (1536 kmers) and workgroups of 128 threads: 0.876-0.895 of
    //     8 TB/s against 0.819-0.840 with round 2's 16 KiB x 256 and 0.866-0.872 with 24 KiB x 256;
    //   * anything else (two plain allocations, one region class): 16 KiB x 256 threads stays: 0.788-0.801 in six processes, where
    //     24 KiB is bimodal (0.727-0.728 in three of them, 0.79-0.81 in the others) and 128 threads lose 3-5 %.
    // Every other shape keeps 16 KiB and 256 threads (C3 0.80-0.82 / 0.72-0.77 with 128, C4 0.87 at 512 x 256 in both
    // placements, C5 0.75-0.76 / 0.72-0.76).
    const bool two_word_streams = n_words == 1 && stride1 && !a.tuples && a.out_a && (a.out_b || a.out_starts) &&
                                  (MODE == MODE_FW || MODE == MODE_CANON);
    const bool spread = two_word_streams && kmers_arena_spread(ctx->arena, a.out_a, a.out_b ? (const void *)a.out_b : (const void *)a.out_starts,
                                                                  (size_t)a.n_kmers * 8u);
    // ONE output array that lies across a class boundary of the arena (kmers_dev_alloc_role(KMERS_ALLOC_LONE_OUTPUT)): written
    // through two windows, one per class (split order), by workgroups of 128 threads -- C3 0.80-0.82 -> 0.867-0.870, a two-word
    // kmer array 0.80 -> 0.856, C5 0.75 -> 0.77; a tuple array (32-byte elements) 0.61 -> 0.82-0.83 with 256 threads and six times
    // the tile (profiles/r03_tuning.md section 5; three fresh processes each)
    const bool materialises = MODE == MODE_FW || MODE == MODE_CANON;
    const uint32_t lone_bytes = a.tuples ? (MODE == MODE_FW ? 16u * n_words : 8u * n_words + 8u) : 8u * n_words;
    const bool lone = materialises && ctx->split_order >= 0 && a.out_a && !a.out_b && !a.out_starts &&
                      kmers_arena_straddles(ctx->arena, a.out_a, (size_t)a.n_kmers * lone_bytes);
    // (strided lone outputs -- C5's SpacedKmers -- want the opposite shape: 256 threads and the longest tile the LDS stream holds,
    // 40 KiB of output per workgroup at J = 3: 0.788-0.804 against 0.765-0.772 at 128 x 16 KiB, tools/r3_c5_fine.sh)
    // Two-word canonical kmers + their hashes (CanonicalDNAMers{33..64} + fx_hash: 24 bytes per kmer in two arrays).  The 16 KiB
    // rule rounds to 256 x 512 = 12 KiB per workgroup here and that is the worst shape measured: 0.69-0.72 in every placement
    // (tools/r3_c2_shapes.sh with LEG=c63h, tools/r3_c63h_plain.sh).  128 threads x 512 kmers: 0.80-0.82 with both arrays in one
    // class, 0.82-0.85 elsewhere; 128 x 768 with the arrays well placed: 0.87-0.88 (and 0.65 in one class: only when the map says so).
    // Four-word kmers + hashes (40 bytes per kmer; the rule gives 256 x 256): 0.58 -> 0.69 at 128 x 512 (0.65 at 256 x 512, 0.68 at
    // 64 x 256; three fresh processes each), where the canonical comparison of four-word kmers is most of what is left; three-word
    // kmers take the same shape unmeasured.
    const bool canon2 = MODE == MODE_CANON && n_words >= 2 && stride1 && !a.tuples && a.out_a && a.out_b;
    const bool canon2_spread = canon2 && n_words == 2 &&
                               kmers_arena_spread(ctx->arena, a.out_a, (size_t)a.n_kmers * 16u, a.out_b, (size_t)a.n_kmers * 8u);
    uint32_t threads = ctx->block_threads > 0 ? (uint32_t)ctx->block_threads
                                              : ((spread || canon2 || (lone && !a.tuples && J == 1)) ? 128u : (uint32_t)BLOCK);
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
    if (ctx->tile_kmers <= 0 && spread) tile = tile * 3 / 2 / pass * pass;  // 24 KiB per workgroup (see above)
    // two-word kmers + their reverse complements (C4) in two well-placed arrays: 12 KiB of each per workgroup instead of 8 --
    // 0.886-0.901 in four fresh processes against 0.867-0.873 (64 threads x 4 KiB run the same; 128 threads lose;
    // tools/r3_c4_shapes.sh); in one class the shorter tile stays (0.86 against 0.81-0.84, profiles/r03_tuning.md section 2)
    // (four-word kmers + reverse complements, 64 bytes per kmer: 256 x 512 = 16 KiB of each array 0.798-0.802 against 0.751-0.756
    // at the rule's 256 x 256; 128 x 512 0.771-0.778; three-word kmers take 512 too, unmeasured)
    if (ctx->tile_kmers <= 0 && n_words >= 2 && stride1 && !a.tuples && MODE == MODE_FW && a.out_a && a.out_b &&
        kmers_arena_spread(ctx->arena, a.out_a, a.out_b, (size_t)a.n_kmers * 8u * (size_t)n_words))
        tile = n_words == 2 ? tile * 3u / 2u / pass * pass : 512u;
    if (ctx->tile_kmers <= 0 && canon2) tile = canon2_spread ? 768u : 512u;      // (see above)
    if (ctx->tile_kmers <= 0 && lone && a.tuples) tile *= 6u;                // tuple arrays through two windows (see above)
    if (ctx->tile_kmers <= 0 && lone && J > 1 && !a.tuples) tile = tile * 5u / 2u;  // strided: 40 KiB per workgroup (clamped below)
    // (round 2 doubled the tile of strided launches -- 32 KiB of output per workgroup; with two lattice kmers per lane the 16 KiB
    // tile is as fast or faster on every box measured in round 3: 0.73-0.76 against 0.70-0.74, profiles/r03_tuning.md)
    tile = std::min<uint32_t>(tile, max_tile_symbols / J);
    tile = std::max<uint32_t>(pass, tile / pass * pass);
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
