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
    //   one-word elements, two arrays (C2 headline, FwRv, | well placed (kmers_arena_spread)     | 128 x 1536 (24 KiB)  | 0.82-0.84 -> 0.88-0.90
    //     kmers + starts)                                 | anything else                        | 256 x 1024 (rule)    | (24 KiB bimodal there)
    //   two-word kmers + reverse complements (C4)         | well placed                          | 256 x 768            | 0.87 -> 0.90-0.915
    //   three- / four-word kmers + reverse complements    | well placed                          | 256 x 512            | 0.75 -> 0.80 (four-word)
    //   two-word canonical kmers + hashes                 | well placed                          | 128 x 768            | 0.71 -> 0.87-0.88
    //                                                     | anything else                        | 128 x 512            | 0.70 -> 0.80-0.85
    //   three- / four-word canonical kmers + hashes       | any                                  | 128 x 512            | 0.58 -> 0.69 (four-word)
    //   ONE output array across a class boundary of the   | kmers_arena_straddles                | 128 x 16 KiB, split  | C3 0.81 -> 0.87, two-word 0.80 -> 0.86
    //     arena (kmers_dev_alloc_role LONE_OUTPUT)        |   ... strided (SpacedKmers, C5)      | 256 x 40 KiB, split  | 0.75 -> 0.78-0.79
    //                                                     |   ... tuple elements                 | 256 x 6 x rule, split| 0.61 -> 0.82-0.83
    //   everything else                                   |                                      | 256 x rule           |
    //
    // "well placed" = both arrays inside the context's arena, in runs whose MEASURED two-stream rate is within 5 % of the block's
    // best pair; "split" = two write windows half an array apart (stream_kernel.hpp, SPLIT ORDER).  KMERS_PARAM_TILE_KMERS /
    // _BLOCK_THREADS / _SPLIT_ORDER override the table (tests, tools/).
    const bool materialises = MODE == MODE_FW || MODE == MODE_CANON;
    const bool two_arrays = materialises && stride1 && a.out_a && !a.tuples;
    const bool one_word_pair = two_arrays && n_words == 1 && (a.out_b || a.out_starts);
    const void *const second = a.out_b ? (const void *)a.out_b : (const void *)a.out_starts;
    const size_t bytes_a = (size_t)a.n_kmers * 8u * (size_t)n_words;
    const size_t bytes_b = (size_t)a.n_kmers * (MODE == MODE_CANON || !a.out_b ? 8u : 8u * (size_t)n_words);
    // Blocks of the class pool (pool_api.hip) are assembled so that the arrays of a launch lie in different region classes at every
    // relative position, and a lone output's second half in another class than its first: the same two questions the arena's map
    // answers for its blocks.
    constexpr float PLACED = 0.9f;
    const bool pool_pair = materialises && a.out_a && second && pool_arrays_differ(ctx, a.out_a, bytes_a, second, bytes_b) >= PLACED;
    const bool in_arena = ctx->shared_arena != nullptr;
    const bool spread = one_word_pair && (pool_pair || kmers_arena_spread(ctx->arena(), a.out_a, second, (size_t)a.n_kmers * 8u));
    const bool fwrc_wide = two_arrays && MODE == MODE_FW && n_words >= 2 && a.out_b &&
                           (pool_pair || kmers_arena_spread(ctx->arena(), a.out_a, a.out_b, bytes_a));
    const bool canon_wide = two_arrays && MODE == MODE_CANON && n_words >= 2 && a.out_b;
    const bool canon2_spread = canon_wide && n_words == 2 &&
                               (pool_pair || kmers_arena_spread(ctx->arena(), a.out_a, (size_t)a.n_kmers * 16u, a.out_b, (size_t)a.n_kmers * 8u));
    const uint32_t lone_bytes = a.tuples ? (MODE == MODE_FW ? 16u * n_words : 8u * n_words + 8u) : 8u * n_words;
    // ONE output array whose two halves lie in different classes (across a class boundary of the arena, or a block of the pool
    // made that way by role) is written through two windows half an array apart
    const bool lone_output = materialises && a.out_a && !a.out_b && !a.out_starts && ctx->split_order >= 0;
    const bool lone_pool = lone_output && pool_halves_differ(ctx, a.out_a, (size_t)a.n_kmers * lone_bytes) >= 0.75f;  // (whole handles: short arrays cannot do better)
    bool lone = lone_pool || (lone_output && kmers_arena_straddles(ctx->arena(), a.out_a, (size_t)a.n_kmers * lone_bytes));  // (the calibration below may overrule it)
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
    // The table is what rounds 3-4 measured on a handful of boxes; the region map of an arena can be finer than an array (runs of one
    // 4 GiB granule), and there the table's shape lost 13 % to the base rule (headline 0.70 instead of 0.80, profiles/r04_shape.md).
    // So the first large SYNCHRONOUS launch into arrays of the ARENA for which the table departs from the rule times both and
    // remembers -- per launch configuration and placement, not per pointer (context.hpp, shape_choice).  Blocks of the pool are
    // placed well by construction and are never timed; nothing is ever timed inside a KMERS_ASYNC call.
    const bool arena_placed = in_arena && !pool_pair && !lone_pool && (spread || fwrc_wide || canon_wide || lone);
    if (arena_placed && ctx->tile_kmers <= 0 && ctx->block_threads <= 0 && ctx->shape_calibrate > 0 && !ctx->calibrating &&
        (uint64_t)a.n_kmers * out_bytes >= ((uint64_t)1 << 30)) {
        const kmers_arena &ar = ctx->arena();
        auto run_at = [&](const void *q) -> uint64_t {
            const char *c = static_cast<const char *>(q);
            if (!q || ar.run_start.empty() || c < ar.base || c >= ar.base + ar.bytes) return 0xffu;
            return (uint64_t)kmers_arena_run_of(ar, (size_t)(c - ar.base)) & 0xffu;
        };
        uint64_t bucket = 0;  // log2 of the bytes written
        for (uint64_t v = (uint64_t)a.n_kmers * out_bytes; v > 1; v >>= 1) ++bucket;
        const uint64_t key = (uint64_t)MODE | (uint64_t)n_words << 4 | (uint64_t)(a.tuples ? 1 : 0) << 8 | (uint64_t)(a.out_b ? 1 : 0) << 9 |
                             (uint64_t)(a.out_starts ? 1 : 0) << 10 | (uint64_t)(src_bits & 15) << 11 | (uint64_t)(dst_bits & 15) << 15 |
                             (uint64_t)std::min<uint32_t>(J, 255u) << 19 | bucket << 27 | run_at(a.out_a) << 35 | run_at(second) << 43;
        const kmers_ctx::shape_choice *hit = nullptr;
        for (const auto &c : ctx->shape_cache)
            if (c.key == key) hit = &c;
        if (!hit && !(ctx->call_flags & KMERS_ASYNC)) {
            const uint32_t rule_pass = ((stride1 || pair) && n_words == 1 ? 2u : 1u) * (uint32_t)BLOCK;
            const int cand[2][2] = {{(int)threads, (int)std::max<uint32_t>(pass, tile / pass * pass)},
                                    {BLOCK, (int)default_tile(out_bytes, rule_pass)}};
            // Timed fairly: the first launches of a process find the device's clocks idle, so both shapes run once untimed and are
            // then timed ALTERNATELY, A B | B A | A B, each launch between its own pair of events; the better of three counts.
            // The table is what several boxes measured: the rule has to beat it by 3 % to overrule it.
            float ms[2] = {1e30f, 1e30f};
            for (auto &e : ctx->cal_events)
                if (!e) HIP_TRY(ctx, hipEventCreate(&e));
            ctx->calibrating = true;
            ++ctx->calibrations;
            int rc = KMERS_OK;
            const int64_t split_saved = ctx->split_order;
            static const int order[8] = {0, 1, 0, 1, 1, 0, 0, 1};  // (the first two: warm-up)
            for (int i = 0; i < 8 && rc == KMERS_OK; ++i) {
                const int c = order[i];
                ctx->block_threads = cand[c][0];
                ctx->tile_kmers = cand[c][1];
                ctx->split_order = (c == 1 && lone) ? -1 : split_saved;  // (the base rule writes a lone output through ONE window)
                if (i >= 2 && hipEventRecord(ctx->cal_events[0], ctx->stream) != hipSuccess) rc = KMERS_E_HIP;
                StreamArgs copy = a;
                if (rc == KMERS_OK) rc = launch_stream<MODE>(ctx, copy, src_bits, dst_bits, n_words, vec_ok, dyn_lds);
                if (i >= 2 && rc == KMERS_OK) {
                    float t = 0.f;
                    if (hipEventRecord(ctx->cal_events[1], ctx->stream) != hipSuccess || hipEventSynchronize(ctx->cal_events[1]) != hipSuccess ||
                        hipEventElapsedTime(&t, ctx->cal_events[0], ctx->cal_events[1]) != hipSuccess)
                        rc = KMERS_E_HIP;
                    else if (t < ms[c]) ms[c] = t;
                }
            }
            ctx->block_threads = 0;
            ctx->tile_kmers = 0;
            ctx->split_order = split_saved;
            ctx->calibrating = false;
            if (rc != KMERS_OK) return rc == KMERS_E_HIP ? fail(ctx, KMERS_E_HIP, "launch-shape calibration") : rc;
            const int best = ms[1] < 0.97f * ms[0] ? 1 : 0;
            if (ctx->shape_cache.size() >= 64) ctx->shape_cache.erase(ctx->shape_cache.begin());
            ctx->shape_cache.push_back({key, cand[best][0], cand[best][1], best == 1, ms[0], ms[1]});
            hit = &ctx->shape_cache.back();
        }
        if (hit) {
            threads = (uint32_t)hit->threads;
            tile = (uint32_t)hit->tile;
            if (hit->rule) lone = false;  // (one write window)
            ctx->last_cal_table_ms = hit->table_ms;
            ctx->last_cal_rule_ms = hit->rule_ms;
            ctx->last_cal_rule = hit->rule ? 1 : 0;
        } else {
            ctx->last_cal_table_ms = ctx->last_cal_rule_ms = 0.f;
            ctx->last_cal_rule = 0;
        }
    } else if (!ctx->calibrating) {
        ctx->last_cal_table_ms = ctx->last_cal_rule_ms = 0.f;
        ctx->last_cal_rule = 0;
    }
    // (strided launches in ONE class: round 2's 32 KiB tile lost to 16 KiB on every box of round 3, 0.70-0.74 against 0.73-0.76)
    const uint32_t pass_now = ((stride1 || pair) && n_words == 1 ? 2u : 1u) * threads;  // (the cache may have changed `threads`)
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
