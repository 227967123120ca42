// consumers_api.hip -- the fused consumers (include/kmers_hip.h): nothing is materialised per kmer.  kmers_reduce_xor(_iter)
// (test/benchmark.jl:9-15, :35-94), kmers_minhash (docs/src/minhash.md:31-35), kmers_minimizers (docs/src/replacements.md:33-51),
// kmers_composition (docs/src/composition.md:28-39).
#include "../../include/kmers_hip.h"

#include "stream_launch.hpp"
#include "wide_kernel.hpp"
#include "wide_tile_kernel.hpp"
#include "composition_kernel.hpp"
#include "run_kernel.hpp"
#include "sketch_prune_kernel.hpp"

using namespace kmers;

namespace {

// Common launch of a fused-consumer mode (nothing materialised per kmer).
template <int MODE>
int launch_fused(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int dst_bits, StreamArgs &a, size_t dyn_lds = 0) {
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = kmers_count(seq->n_bases, k, 1);
    a.inspect_end = seq->n_bases;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = 1;
    a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
    int64_t saved = ctx->max_grid;
    if (ctx->max_grid <= 0) ctx->max_grid = 256 * 8;  // persistent grid: no output stream to pace
    int rc = launch_stream<MODE>(ctx, a, seq->src_bits, dst_bits, kmers_words_per_kmer(k, dst_bits), true, dyn_lds);
    ctx->max_grid = saved;
    return rc;
}

// Kmers of more than four words (and strides no tile can stage): one lane per kmer, the width a run-time argument
// (wide_kernel.hpp).  `a` carries the consumer's own fields; the sequence fields are filled here.
template <int CMODE>
int launch_wide_consumer(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int stride, int dst_bits, StreamArgs &a) {
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = kmers_count(seq->n_bases, k, stride);
    a.inspect_end = seq->n_bases;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = (uint32_t)stride;
    a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
    const uint32_t nwu = (uint32_t)kmers_words_per_kmer(k, dst_bits);
    if (!ctx->wide_no_tiles) {  // stride 1: the tile form (wide_tile_kernel.hpp)
        const int tiled = launch_wide_tile<CMODE == WIDE_XOR ? WMODE_XOR : WMODE_SKETCH>(ctx, a, seq->src_bits, dst_bits, nwu);
        if (tiled >= 0) return tiled;
    }
    dim3 grid((unsigned)((a.n_kmers + BLOCK - 1) / BLOCK)), block(BLOCK);
#define WIDEC(SB, DB) hipLaunchKernelGGL((wide_consumer_kernel<SB, DB, CMODE>), grid, block, 0, ctx->stream, a, nwu)
    KMERS_WIDE_DISPATCH(WIDEC, seq->src_bits, dst_bits);
#undef WIDEC
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

// Fused consumers of one- and two-word 2-bit kmers (K <= 64): the rolling run kernel (run_kernel.hpp);
// everything else (three- and four-word kmers, 4-bit kmer alphabets) goes through the stream kernel's fused modes.
template <int RMODE, int SMODE>
int launch_consumer(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int dst_bits, StreamArgs &a, size_t best_bytes = 0) {
    if (kmers_words_per_kmer(k, dst_bits) > 4)
        return launch_wide_consumer<RMODE == RMODE_XOR ? WIDE_XOR : WIDE_SKETCH>(ctx, seq, st, k, 1, dst_bits, a);
    if (dst_bits != 2 || k > 64) return launch_fused<SMODE>(ctx, seq, st, k, dst_bits, a);
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = kmers_count(seq->n_bases, k, 1);
    a.inspect_end = seq->n_bases;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = 1;
    a.ascii_table = ascii_table(ctx, 2, seq->alphabet);
    a.n_tiles = (a.n_kmers + RTILE - 1) / RTILE;
    a.stamps = reinterpret_cast<uint64_t *>(ctx->stamps_ptr);  // (diagnostic builds only)
    const uint64_t resident = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (uint64_t)ctx->n_cus * 8;  // persistent grid
    dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, resident)), block(RBLOCK);
#define RUNK(SB)                                                                                             \
    do {                                                                                                     \
        if (RMODE == RMODE_XOR && !a.xor_canonical) {                                                        \
            if (k <= 16) hipLaunchKernelGGL((run_kernel<SB, RMODE_XOR, 1, false, true>), grid, block, best_bytes, ctx->stream, a); \
            else if (k <= 32) hipLaunchKernelGGL((run_kernel<SB, RMODE_XOR, 1, false>), grid, block, best_bytes, ctx->stream, a);  \
            else hipLaunchKernelGGL((run_kernel<SB, RMODE_XOR, 2, false>), grid, block, best_bytes, ctx->stream, a);         \
        } else if (k <= 16) hipLaunchKernelGGL((run_kernel<SB, RMODE, 1, true, true>), grid, block, best_bytes, ctx->stream, a); \
        else if (k <= 32) hipLaunchKernelGGL((run_kernel<SB, RMODE, 1, true>), grid, block, best_bytes, ctx->stream, a);   \
        else hipLaunchKernelGGL((run_kernel<SB, RMODE, 2, true>), grid, block, best_bytes, ctx->stream, a);  \
    } while (0)
    if (seq->src_bits == 8) RUNK(8);
    else if (seq->src_bits == 4) RUNK(4);
    else RUNK(2);
#undef RUNK
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

}  // namespace

extern "C" {

int kmers_reduce_xor(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, int canonical, uint64_t *out_value,
                     int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, 1, dst_bits, flags & ~KMERS_ASYNC)) {
        if (res) res->status = rc;
        return rc;
    }
    if (!out_value) return fail(ctx, KMERS_E_BADARG, "out_value is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out_value = 0;
    const uint64_t n = kmers_count(seq->n_bases, k, 1);
    if (n == 0) return KMERS_OK;
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, 8, ctx->stream));
    StreamArgs a{};
    a.out_a = ctx->d_scratch;
    a.xor_canonical = canonical ? 1u : 0u;
    if (int rc = launch_consumer<RMODE_XOR, MODE_XOR>(ctx, seq, st, k, dst_bits, a)) return rc;
    return collect(ctx, res, n, out_value);
}

int kmers_reduce_xor_iter(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, int iter, int stride, uint64_t *out_value,
                          int flags, kmers_result *res) {
    if (iter == KMERS_ITER_FW || iter == KMERS_ITER_CANONICAL)
        return kmers_reduce_xor(ctx, seq, k, dst_bits, iter == KMERS_ITER_CANONICAL, out_value, flags, res);
    clear(res);
    if (iter != KMERS_ITER_SPACED && iter != KMERS_ITER_UNAMBIGUOUS) return ctx ? fail(ctx, KMERS_E_BADARG, "unknown iterator") : KMERS_E_BADARG;
    if (iter == KMERS_ITER_UNAMBIGUOUS) dst_bits = 2;
    if (int rc = check_common(ctx, seq, k, stride, dst_bits, flags & ~KMERS_ASYNC)) {
        if (res) res->status = rc;
        return rc;
    }
    if (!out_value) return fail(ctx, KMERS_E_BADARG, "out_value is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *out_value = 0;
    const int nw = kmers_words_per_kmer(k, dst_bits);
    if (iter == KMERS_ITER_SPACED) {
        const uint64_t n = kmers_count(seq->n_bases, k, stride);
        if (n == 0) return KMERS_OK;
        Staged st;
        if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, 8, ctx->stream));
        StreamArgs a{};
        if (nw > 4 || (uint64_t)stride * (uint64_t)dst_bits > 64) {
            // kmers of more than four words, or windows so far apart that a tile would stage mostly unread symbols
            // (SpacedKmers.jl:121-139 with J >= K never inspects the gaps): one lane per kmer
            a.out_a = ctx->d_scratch;
            a.xor_canonical = 0;
            if (int rc = launch_wide_consumer<WIDE_XOR>(ctx, seq, st, k, stride, dst_bits, a)) return rc;
            return collect(ctx, res, n, out_value);
        }
        a.src = st.d_words;
        a.first_bit = st.first_bit;
        a.n_bases = seq->n_bases;
        a.n_kmers = n;
        a.inspect_end = (n - 1) * (uint64_t)stride + (uint64_t)k;
        a.out_a = ctx->d_scratch;
        a.err_slot = ctx->d_err;
        a.err_origin = seq->index_origin;
        a.k = (uint32_t)k;
        a.stride = (uint32_t)stride;
        a.xor_canonical = 0;
        a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
        int64_t saved = ctx->max_grid;
        if (ctx->max_grid <= 0) ctx->max_grid = (int64_t)ctx->n_cus * 8;  // persistent grid: nothing is streamed out
        const int rc = launch_stream<MODE_XOR>(ctx, a, seq->src_bits, dst_bits, nw, true);
        ctx->max_grid = saved;
        if (rc) return rc;
        return collect(ctx, res, n, out_value);
    }
    return unambiguous_xor(ctx, seq, k, stride, out_value, flags, res);
}

static int minhash_impl(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t seed, uint64_t s,
                        uint64_t *out_hashes, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, 1, dst_bits, flags & ~KMERS_ASYNC)) {
        if (res) res->status = rc;
        return rc;
    }
    if (flags & KMERS_ASYNC) return fail(ctx, KMERS_E_BADARG, "kmers_minhash is synchronous");
    if (s == 0 || !out_hashes) return fail(ctx, KMERS_E_BADARG, "sketch size must be positive and out_hashes non-NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = kmers_count(seq->n_bases, k, 1);
    if (n == 0) return KMERS_OK;
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;

    constexpr uint32_t RECENT_SLOTS = 1u << 16;
    if (!ctx->d_recent) {
        hipError_t e = dev_malloc(ctx, reinterpret_cast<void **>(&ctx->d_recent), (size_t)RECENT_SLOTS * 8);
        if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc(recent candidates)", e);
    }
    // (a sequence that fits one round needs no duplicate filter: everything is a candidate once and the
    // prune kernel deduplicates)
    const bool one_round = n <= SKETCH_LDS_VALUES - 4096;  // (also below the smallest candidate buffer)
    if (!one_round) HIP_TRY(ctx, hipMemsetAsync(ctx->d_recent, 0xFF, (size_t)RECENT_SLOTS * 8, ctx->stream));

    // ---- device-resident path (s <= 4096): the threshold and the running bottom-s set stay in HBM, a
    // one-workgroup bitonic sort/unique kernel prunes between chunks, every round is enqueued without a
    // host round trip.  Chunk r+1 is `ratio` times everything before it, which yields about ratio*s
    // candidates (half the buffer); an adversarial order can overflow it -> flag -> feedback path below.
    if (s <= 4096 && !ctx->sketch_host_only) {
        // candidates per round: the prune kernel's pivot cut needs only ~1.5 s values in LDS, so the buffer
        // can be much larger than the LDS sort (fewer, longer rounds); for the largest sketches the cut does
        // not fit and everything must (12288 + 4096 <= the LDS sort)
        const bool cut_fits = 1.25 * (1.5 * (double)s + 8.0 * std::sqrt((double)s) + 32.0) <= (double)SKETCH_LDS_VALUES / 2;
        const uint64_t dcap = cut_fits ? 65536 : SKETCH_LDS_VALUES - 4096;
        // new candidates in a chunk of ratio*done kmers ~ ratio * Gamma(s): keep the buffer at mean + a wide
        // margin (relative spread 1/sqrt(s); s = 1 needs ~18x for a 1e-8 overflow probability)
        const double margin = 2.0 + 16.0 / std::sqrt((double)s);
        const uint64_t ratio = std::max<uint64_t>(1, (uint64_t)((double)dcap / ((double)s * margin)));
        if (int rc = ensure_stage(ctx, 3, (size_t)(dcap + 4096 + 8) * 8)) return rc;
        uint64_t *d_cand = static_cast<uint64_t *>(ctx->stage[3]);
        uint64_t *d_best = d_cand + dcap;
        uint64_t *d_state = d_best + 4096;                              // [n_best, threshold, overflow, counter]
        HIP_TRY(ctx, hipMemsetAsync(d_state, 0, 32, ctx->stream));       // {0, ~0, 0, 0} without a pageable H2D copy
        HIP_TRY(ctx, hipMemsetAsync(d_state + 1, 0xFF, 8, ctx->stream));
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(sketch_prune_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SKETCH_LDS_VALUES * 8));
        auto launch_chunk = [&](uint64_t done, uint64_t m) -> int {
            kmers_seq view = *seq;
            Staged vst = st;
            vst.first_bit = st.first_bit + done * (uint64_t)seq->src_bits;
            view.n_bases = m + (uint64_t)k - 1;
            view.index_origin = seq->index_origin + done;
            StreamArgs a{};
            a.out_a = d_cand;
            a.out_b = d_state + 3;
            a.seed = seed;
            a.threshold_ptr = d_state + 1;
            a.best = d_best;
            a.best_n_ptr = d_state;
            a.recent = one_round ? nullptr : ctx->d_recent;
            a.recent_mask = RECENT_SLOTS - 1;
            a.capacity = dcap;
            if (int rc = launch_consumer<RMODE_SKETCH, MODE_SKETCH>(ctx, &view, vst, k, dst_bits, a, (size_t)s * 8)) return rc;
            hipLaunchKernelGGL(sketch_prune_kernel, dim3(1), dim3(1024), SKETCH_LDS_VALUES * 8, ctx->stream, d_best, d_state,
                               d_cand, dcap, (uint32_t)s);
            HIP_TRY(ctx, hipGetLastError());
            return KMERS_OK;
        };
        uint64_t *h_state = reinterpret_cast<uint64_t *>(ctx->h_bounce), *h_best = h_state + 8;
        // ---- single sweep with a provisional threshold: hashes are close to uniform, so the
        // (1.5 s + slack) / n quantile of the 64-bit range should leave about 1.5 s candidates from the WHOLE
        // sequence (2.5 s with the factor below) -- one candidate kernel and one merge instead of geometric rounds.  If at least s distinct
        // values lie below it they are the sketch; otherwise (skewed or heavily duplicated hashes, or a
        // sequence with fewer than s distinct kmers) the rounds below start from scratch.
        // (2.5 s rather than 1.5 s: in repeat-rich sequence half of the kmers below the threshold may be duplicates)
        const double frac = (2.5 * (double)s + 8.0 * std::sqrt((double)s) + 32.0) / (double)n;
        // (sketches too large for the pivot cut have the small buffer: 2.5 s + slack must still fit it with room to spare)
        const bool sweep_fits = cut_fits || 2.5 * (double)s + 8.0 * std::sqrt((double)s) + 32.0 + 1024.0 <= (double)dcap;
        if (sweep_fits && !one_round && frac < 0.25) {
            // (not through h_bounce: a short host source may still be on its way to HBM from there)
            uint64_t *h_up = ctx->h_result + 4;  // pinned words 4..7: {n_best, threshold, overflow, counter}
            h_up[0] = 0;
            h_up[1] = (uint64_t)(frac * 18446744073709551616.0);
            h_up[2] = h_up[3] = 0;
            HIP_TRY(ctx, hipMemcpyAsync(d_state, h_up, 32, hipMemcpyHostToDevice, ctx->stream));
            if (int rc = launch_chunk(0, n)) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(h_state, d_state, 32, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(h_state + 4, ctx->d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(h_best, d_best, (size_t)s * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
#ifdef KMERS_SKETCH_DEBUG
            std::fprintf(stderr, "provisional sweep: n_best %llu threshold %llx overflow %llu counter %llu err %llx T0 %llx\n",
                         (unsigned long long)h_state[0], (unsigned long long)h_state[1], (unsigned long long)h_state[2],
                         (unsigned long long)h_state[3], (unsigned long long)h_state[4], (unsigned long long)h_up[1]);
#endif
            if (h_state[4] == NO_ERROR_POS && h_state[2] == 0 && h_state[0] == s) {
                std::memcpy(out_hashes, h_best, (size_t)s * 8);
                if (res) { res->status = KMERS_OK; res->n_out = s; }
                return KMERS_OK;
            }
            // not enough below the provisional threshold (or an EncodeError, handled by the paths below): start over
            HIP_TRY(ctx, hipMemsetAsync(d_state, 0, 32, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(d_state + 1, 0xFF, 8, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_recent, 0xFF, (size_t)RECENT_SLOTS * 8, ctx->stream));
        }
        uint64_t done = 0, chunk = std::min<uint64_t>(n, dcap);         // first chunk: everything is a candidate
        while (done < n) {
            const uint64_t m = std::min<uint64_t>(chunk, n - done);
            if (int rc = launch_chunk(done, m)) return rc;
            done += m;
            chunk = std::max<uint64_t>(dcap / 2, ratio * done);
        }
        // results through pinned memory, one wait: state, the error slot, and (optimistically) the sketch
        HIP_TRY(ctx, hipMemcpyAsync(h_state, d_state, 32, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(h_state + 4, ctx->d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(h_best, d_best, (size_t)s * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        // an EncodeError anywhere in the sequence: report the first one (positions are relative to a chunk,
        // so re-run the feedback path, which attributes it exactly)
        const unsigned long long epos = h_state[4];
        if (epos == NO_ERROR_POS && h_state[2] == 0) {
            const uint64_t nb = h_state[0];
            if (nb) std::memcpy(out_hashes, h_best, nb * 8);
            if (res) { res->status = KMERS_OK; res->n_out = nb; }
            return KMERS_OK;
        }
        if (epos != NO_ERROR_POS) {  // re-arm the slot; the feedback path below finds and reports the error
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
    }

    // Candidate buffer in HBM; the host keeps the running bottom-s set (a few thousand values).
    // (The table of recent candidates starts empty: a device-path attempt may have left entries behind.)
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_recent, 0xFF, (size_t)RECENT_SLOTS * 8, ctx->stream));
    const uint64_t cap = std::max<uint64_t>((uint64_t)1 << 16, 8 * s);
    if (int rc = ensure_stage(ctx, 3, (size_t)(cap + 2) * 8)) return rc;
    uint64_t *d_cand = static_cast<uint64_t *>(ctx->stage[3]);
    uint64_t *d_counter = d_cand + cap;
    std::vector<uint64_t> best, chunk_vals;
    uint64_t threshold = ~0ull;  // hashes strictly below it are candidates
    uint64_t done = 0;
    // One sweep with a provisional threshold first (see the device-resident path): about 2.5 s candidates from the
    // whole sequence, sorted on the host; accepted if they hold at least s distinct values.
    {
        const double frac = (2.5 * (double)s + 8.0 * std::sqrt((double)s) + 32.0) / (double)n;
        if (!ctx->sketch_host_only && frac < 0.25 && 3.0 * (double)s + 1024.0 < (double)cap) {
            HIP_TRY(ctx, hipMemsetAsync(d_counter, 0, 8, ctx->stream));
            StreamArgs a{};
            a.out_a = d_cand;
            a.out_b = d_counter;
            a.seed = seed;
            a.threshold = (uint64_t)(frac * 18446744073709551616.0);
            a.capacity = cap;
            a.recent = ctx->d_recent;
            a.recent_mask = RECENT_SLOTS - 1;
            if (int rc = launch_consumer<RMODE_SKETCH, MODE_SKETCH>(ctx, seq, st, k, dst_bits, a)) return rc;
            uint64_t *h = reinterpret_cast<uint64_t *>(ctx->h_bounce);
            HIP_TRY(ctx, hipMemcpyAsync(h, d_counter, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(h + 1, ctx->d_err, 8, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            const uint64_t count = h[0];
            if (h[1] == NO_ERROR_POS && count <= cap) {
                best.resize(count);
                if (count) HIP_TRY(ctx, hipMemcpy(best.data(), d_cand, count * 8, hipMemcpyDeviceToHost));
                std::sort(best.begin(), best.end());
                best.erase(std::unique(best.begin(), best.end()), best.end());
                if (best.size() >= s) {
                    std::memcpy(out_hashes, best.data(), (size_t)s * 8);
                    if (res) { res->status = KMERS_OK; res->n_out = s; }
                    return KMERS_OK;
                }
            }
            // not enough below the provisional threshold, or an EncodeError (the rounds below attribute it): start over
            best.clear();
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_recent, 0xFF, (size_t)RECENT_SLOTS * 8, ctx->stream));
        }
    }
    // Geometric chunks: with the threshold at the s-th smallest value seen so far, a chunk r times
    // as long as everything before it yields about r*s new candidates, so the buffer stays small.
    uint64_t chunk = std::min<uint64_t>(n, cap / 2);
    while (done < n) {
        uint64_t m = std::min<uint64_t>(chunk, n - done);
        kmers_seq view = *seq;
        Staged vst = st;
        vst.first_bit = st.first_bit + done * (uint64_t)seq->src_bits;
        view.n_bases = m + (uint64_t)k - 1;
        HIP_TRY(ctx, hipMemsetAsync(d_counter, 0, 8, ctx->stream));
        view.index_origin = seq->index_origin + done;  // error positions of this launch are relative to the chunk
        StreamArgs a{};
        a.out_a = d_cand;
        a.out_b = d_counter;
        a.seed = seed;
        a.threshold = threshold;
        a.capacity = cap;
        a.recent = ctx->d_recent;
        a.recent_mask = RECENT_SLOTS - 1;
        if (int rc = launch_consumer<RMODE_SKETCH, MODE_SKETCH>(ctx, &view, vst, k, dst_bits, a)) return rc;
        uint64_t count = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&count, d_counter, 8, hipMemcpyDeviceToHost, ctx->stream));
        // chunks run in sequence order, so the first chunk that reports an EncodeError holds the
        // first offending symbol of the whole sequence
        if (int erc = collect(ctx, res, 0)) return erc;
        const uint64_t got = std::min<uint64_t>(count, cap);
        chunk_vals.resize(got);
        if (got) HIP_TRY(ctx, hipMemcpy(chunk_vals.data(), d_cand, got * 8, hipMemcpyDeviceToHost));
        best.insert(best.end(), chunk_vals.begin(), chunk_vals.end());
        std::sort(best.begin(), best.end());
        best.erase(std::unique(best.begin(), best.end()), best.end());
        if (best.size() > s) best.resize(s);
        if (best.size() == s) threshold = best.back();  // only values below the current s-th smallest matter
        if (count > cap) {
            // buffer overflow (adversarial order): the threshold just tightened, redo this chunk.  Dropped
            // candidates are in the table of recent ones but nowhere else: forget them.
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_recent, 0xFF, (size_t)RECENT_SLOTS * 8, ctx->stream));
            if (m > 1) chunk = std::max<uint64_t>(1, m / 2);
            continue;
        }
        done += m;
        chunk = std::max<uint64_t>(chunk, 3 * done);  // next chunk 3x everything so far: about 3s candidates (< cap)
    }
    std::memcpy(out_hashes, best.data(), best.size() * 8);
    if (res) res->n_out = best.size();
    return KMERS_OK;
}

int kmers_minimizers(kmers_ctx *ctx, const kmers_seq *seq, int k, int w, int stride, int dst_bits, int mode,
                     uint64_t *out_kmers, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, stride, dst_bits, flags)) {
        if (res) res->status = rc;
        return rc;
    }
    if (w < 1 || w > 4096 || (mode != 0 && mode != 1)) return fail(ctx, KMERS_E_BADARG, "window must be 1..4096 kmers, mode 0 or 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nw = kmers_words_per_kmer(k, dst_bits);
    const uint64_t span = (uint64_t)k + (uint64_t)w - 1;
    const uint64_t n = seq->n_bases < span ? 0 : (seq->n_bases - span) / (uint64_t)stride + 1;
    if (n == 0) return KMERS_OK;
    if (!out_kmers) return fail(ctx, KMERS_E_BADARG, "out_kmers is NULL");
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
    const bool dev = flags & KMERS_MEM_DEVICE;
    uint64_t *d_out = out_kmers;
    const size_t bytes = (size_t)n * nw * 8;
    if (!dev) {
        if (int rc = ensure_stage(ctx, 1, bytes)) return rc;
        d_out = static_cast<uint64_t *>(ctx->stage[1]);
    }
    if ((nw == 2 || nw == 4) && !aligned16(d_out)) return fail(ctx, KMERS_E_BADARG, "two- and four-word kmer outputs must be 16-byte aligned");
    StreamArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = n;
    a.inspect_end = (n - 1) * (uint64_t)stride + span;  // every symbol of every window is read
    a.out_a = d_out;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = (uint32_t)stride;
    a.window_kmers = (uint32_t)w;
    a.minimizer_mode = (uint32_t)mode;
    a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
    if (nw > 4 || (uint64_t)stride * (uint64_t)dst_bits > 64 * 8) {
        // kmers of more than four words / windows further apart than a tile stages: one lane per window (wide_kernel.hpp)
        const uint32_t nwu = (uint32_t)nw;
        dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
#define WIDEM(SB, DB) hipLaunchKernelGGL((wide_minimizer_kernel<SB, DB>), grid, block, 0, ctx->stream, a, nwu)
        KMERS_WIDE_DISPATCH(WIDEM, seq->src_bits, dst_bits);
#undef WIDEM
        HIP_TRY(ctx, hipGetLastError());
    } else if (int rc = launch_stream<MODE_MINIMIZER>(ctx, a, seq->src_bits, dst_bits, nw, false)) {
        // strides >= span leave gaps the reference's loop never reads: the validation is restricted to the windows
        return rc;
    }
    if (flags & KMERS_ASYNC) {
        if (res) { res->status = KMERS_OK; res->n_out = n; }
        return KMERS_OK;
    }
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(out_kmers, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return collect(ctx, res, n);
}

int kmers_composition(kmers_ctx *ctx, const kmers_seq *seq, int k, uint32_t *out_counts, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, 1, 2, flags & ~KMERS_ASYNC)) {
        if (res) res->status = rc;
        return rc;
    }
    if (k > 16) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_composition supports K <= 16 (4^K uint32 counters: 17 GB at K = 16)");
    if (!out_counts) return fail(ctx, KMERS_E_BADARG, "out_counts is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bins = (size_t)1 << (2 * k);
    const bool dev = flags & KMERS_MEM_DEVICE;
    uint32_t *d_counts = out_counts;
    if (!dev) {
        if (int rc = ensure_stage(ctx, 1, bins * 4)) return rc;
        d_counts = static_cast<uint32_t *>(ctx->stage[1]);
    }
    HIP_TRY(ctx, hipMemsetAsync(d_counts, 0, bins * 4, ctx->stream));
    const uint64_t n = kmers_count(seq->n_bases, k, 1);
    if (n) {
        Staged st;
        if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
        if (k <= 10) {
            // private 16-bit histograms in LDS, 65 536 bins per pass (composition_kernel.hpp):
            // 0.4-0.6 ms per Gbase up to K = 8, 0.75 ms per pass beyond (K = 9: 4 passes, K = 10: 16)
            CompositionArgs a{};
            a.src = st.d_words;
            a.first_bit = st.first_bit;
            a.n_bases = seq->n_bases;
            a.n_kmers = n;
            a.n_tiles = (n + CTILE - 1) / CTILE;
            a.counts = d_counts;
            a.err_slot = ctx->d_err;
            a.err_origin = seq->index_origin;
            a.ascii_table = ascii_table(ctx, 2, seq->alphabet);
            a.k = (uint32_t)k;
            a.hist_words = (uint32_t)std::min<size_t>(bins, (size_t)1 << CBINS_LOG2) / 2;
            const uint32_t passes = (uint32_t)std::max<size_t>(1, bins >> CBINS_LOG2);
            const size_t dyn = (size_t)a.hist_words * 4;
            const unsigned per_cu = dyn <= 64 * 1024 ? 2u : 1u;  // 1024-thread workgroups: at most two per CU
            uint64_t resident = (uint64_t)ctx->n_cus * per_cu;
            if (ctx->max_grid > 0) resident = std::min<uint64_t>(resident, (uint64_t)ctx->max_grid);
            dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, resident)), block(CBLOCK);
            auto kern = seq->src_bits == 8 ? composition_kernel<8> : (seq->src_bits == 4 ? composition_kernel<4> : composition_kernel<2>);
            if (dyn > 48 * 1024)
                HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            for (uint32_t p = 0; p < passes; ++p) {
                a.pass = p;
                hipLaunchKernelGGL(kern, grid, block, dyn, ctx->stream, a);
            }
            HIP_TRY(ctx, hipGetLastError());
        } else {
            // 4^11 .. 4^16 counters (16 MiB .. 16 GiB of HBM): 64+ passes would cost more than memory-side global atomics
            // (37 ms per Gbase at K = 12)
            StreamArgs a{};
            a.out_a = reinterpret_cast<uint64_t *>(d_counts);
            if (int rc = launch_fused<MODE_COUNT>(ctx, seq, st, k, 2, a, 0)) return rc;
        }
    }
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(out_counts, d_counts, bins * 4, hipMemcpyDeviceToHost, ctx->stream));
    return collect(ctx, res, n);
}

// The two entry points that allocate host memory (std::vector): no C++ exception may cross the C ABI.
int kmers_minhash(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t seed, uint64_t s,
                  uint64_t *out_hashes, int flags, kmers_result *res) {
    try {
        return minhash_impl(ctx, seq, k, dst_bits, seed, s, out_hashes, flags, res);
    } catch (const std::bad_alloc &) {
        return fail(ctx, KMERS_E_NOMEM, "host allocation failed in kmers_minhash");
    } catch (...) {
        return fail(ctx, KMERS_E_HIP, "unexpected exception in kmers_minhash");
    }
}

}  // extern "C"
