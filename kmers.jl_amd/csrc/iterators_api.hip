// iterators_api.hip -- kmers_fw / kmers_canonical / kmers_spaced (include/kmers_hip.h): FwKmers, FwRvIterator, CanonicalKmers (+ fx_hash)
// and SpacedKmers collected (src/iterators/FwKmers.jl:57-115, CanonicalKmers.jl:54-144, :199-225, SpacedKmers.jl:83-139).
#include "../../include/kmers_hip.h"

#include "stream_launch.hpp"
#include "wide_kernel.hpp"
#include "wide_tile_kernel.hpp"

using namespace kmers;

namespace {

// stride > what a tile can stage: one lane per kmer
template <int SB, int DB>
void launch_gather(int n_words, dim3 grid, dim3 block, hipStream_t st, const StreamArgs &a) {
    switch (n_words) {
        case 1: hipLaunchKernelGGL((gather_kernel<SB, DB, 1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((gather_kernel<SB, DB, 2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((gather_kernel<SB, DB, 3>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((gather_kernel<SB, DB, 4>), grid, block, 0, st, a); break;
    }
}

// Shared body of kmers_fw / kmers_canonical / kmers_spaced.
int run_stream(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits, int mode,
               uint64_t *out_a, uint64_t *out_b, bool b_is_hash, uint64_t seed, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, stride, dst_bits, flags)) {
        if (res) res->status = rc;
        return rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nw = kmers_words_per_kmer(k, dst_bits);
    const uint64_t n = kmers_count(seq->n_bases, k, stride);
    if (n == 0) {  // length(seq) < K: empty iteration, nothing inspected (FwKmers.jl:63)
        if (res) res->status = KMERS_OK;
        return KMERS_OK;
    }
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;

    const bool dev = flags & KMERS_MEM_DEVICE;
    const bool tuples = (flags & KMERS_OUT_TUPLES) != 0;
    if (tuples) {
        if (out_b || !out_a) return fail(ctx, KMERS_E_BADARG, "KMERS_OUT_TUPLES: one interleaved output in the first pointer, second must be NULL");
        if (stride != 1) return fail(ctx, KMERS_E_BADARG, "KMERS_OUT_TUPLES applies to kmers_fw / kmers_canonical / kmers_unambiguous");
    }
    uint64_t *d_a = out_a, *d_b = out_b;
    const size_t tuple_words = mode == MODE_FW ? 2 * (size_t)nw : (size_t)nw + 1;
    const size_t bytes_a = (size_t)n * (tuples ? tuple_words : (size_t)nw) * 8, bytes_b = (size_t)n * (b_is_hash ? 1 : nw) * 8;
    if (!dev) {
        if (out_a) { if (int rc = ensure_stage(ctx, 1, bytes_a)) return rc; d_a = (uint64_t *)ctx->stage[1]; }
        if (out_b) { if (int rc = ensure_stage(ctx, 2, bytes_b)) return rc; d_b = (uint64_t *)ctx->stage[2]; }
    }
    if ((nw == 2 || nw == 4) && ((d_a && !aligned16(d_a)) || (d_b && !b_is_hash && !aligned16(d_b))))
        return fail(ctx, KMERS_E_BADARG, "two- and four-word kmer outputs must be 16-byte aligned");
    if (tuples && !aligned16(d_a)) return fail(ctx, KMERS_E_BADARG, "tuple outputs must be 16-byte aligned");

    StreamArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = n;
    a.inspect_end = (n - 1) * (uint64_t)stride + (uint64_t)k;  // end of the last kmer (== n_bases for stride 1)
    a.out_a = d_a;
    a.out_b = d_b;
    a.seed = seed;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = (uint32_t)stride;
    a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
    a.tuples = tuples ? 1u : 0u;

    int rc;
    if (nw <= 4 && ctx->wide_no_tiles == 2 && !(flags & KMERS_OUT_TUPLES) &&
        (rc = mode == MODE_FW ? launch_wide_tile<WMODE_FW>(ctx, a, seq->src_bits, dst_bits, (uint32_t)nw)
                              : launch_wide_tile<WMODE_CANON>(ctx, a, seq->src_bits, dst_bits, (uint32_t)nw)) >= 0) {
        // A/B only (KMERS_PARAM_WIDE_NO_TILES = 2): kmers of one to four words through the run-time-width tile form
        if (rc) return rc;
    } else if (nw > 4) {
        // kmers of more than four words: at stride 1 the tile form (wide_tile_kernel.hpp: the symbols staged once in LDS, one
        // lane per output word), else the run-time-width kernel that reads single symbols (wide_kernel.hpp), one lane per kmer
        const uint32_t nwu = (uint32_t)nw;
        int tiled = ctx->wide_no_tiles ? -1 : (mode == MODE_FW ? launch_wide_tile<WMODE_FW>(ctx, a, seq->src_bits, dst_bits, nwu)
                                                                : launch_wide_tile<WMODE_CANON>(ctx, a, seq->src_bits, dst_bits, nwu));
        if (tiled > 0) return tiled;
        dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
        if (tiled < 0) {
#define WIDE(SB, DB)                                                                                         \
    do {                                                                                                     \
        if (mode == MODE_FW) hipLaunchKernelGGL((wide_kernel<SB, DB, MODE_FW>), grid, block, 0, ctx->stream, a, nwu);   \
        else hipLaunchKernelGGL((wide_kernel<SB, DB, MODE_CANON>), grid, block, 0, ctx->stream, a, nwu);                \
    } while (0)
        KMERS_WIDE_DISPATCH(WIDE, seq->src_bits, dst_bits);
#undef WIDE
        }
        HIP_TRY(ctx, hipGetLastError());
        rc = KMERS_OK;
    } else if ((uint64_t)stride * (uint64_t)dst_bits > 64 && !ctx->wide_no_tiles && mode == MODE_FW &&
               (rc = launch_wide_tile<WMODE_FW>(ctx, a, seq->src_bits, dst_bits, (uint32_t)nw)) >= 0) {
        // windows further apart than a tile of the stream kernel stages (SpacedKmers{A,K,K} with K > 32 ...): the tile form of
        // the run-time-width kernel, which stages whole stretches and cuts the windows out of LDS (wide_tile_kernel.hpp)
        if (rc) return rc;
    } else if ((uint64_t)stride * (uint64_t)dst_bits > 64) {
        // gather path (forward kmers only: kmers_spaced): one lane per kmer, symbol by symbol
        dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
        if (seq->src_bits == 8 && dst_bits == 2) launch_gather<8, 2>(nw, grid, block, ctx->stream, a);
        else if (seq->src_bits == 8) launch_gather<8, 4>(nw, grid, block, ctx->stream, a);
        else if (seq->src_bits == 4 && dst_bits == 2) launch_gather<4, 2>(nw, grid, block, ctx->stream, a);
        else if (seq->src_bits == 2 && dst_bits == 2) launch_gather<2, 2>(nw, grid, block, ctx->stream, a);
        else if (seq->src_bits == 4 && dst_bits == 4) launch_gather<4, 4>(nw, grid, block, ctx->stream, a);
        else launch_gather<2, 4>(nw, grid, block, ctx->stream, a);
        HIP_TRY(ctx, hipGetLastError());
        rc = KMERS_OK;
    } else {
        const bool vec_ok = (!d_a || aligned16(d_a)) && (!d_b || aligned16(d_b));
        rc = mode == MODE_FW ? launch_stream<MODE_FW>(ctx, a, seq->src_bits, dst_bits, nw, vec_ok)
                             : launch_stream<MODE_CANON>(ctx, a, seq->src_bits, dst_bits, nw, vec_ok);
    }
    if (rc) return rc;
    if (flags & KMERS_ASYNC) {
        if (res) { res->status = KMERS_OK; res->n_out = n; }
        return KMERS_OK;
    }
    const size_t need_a = out_a ? bytes_a : 0, need_b = out_b ? bytes_b : 0;
    const bool bounce = !dev && need_a + need_b <= BOUNCE_OUT;
    char *h_a = ctx->h_bounce + BOUNCE_IN, *h_b = h_a + need_a;
    if (!dev) {
        if (out_a) HIP_TRY(ctx, hipMemcpyAsync(bounce ? (void *)h_a : (void *)out_a, d_a, bytes_a, hipMemcpyDeviceToHost, ctx->stream));
        if (out_b) HIP_TRY(ctx, hipMemcpyAsync(bounce ? (void *)h_b : (void *)out_b, d_b, bytes_b, hipMemcpyDeviceToHost, ctx->stream));
    }
    rc = collect(ctx, res, n);
    if (rc == KMERS_OK && bounce) {
        if (out_a) std::memcpy(out_a, h_a, bytes_a);
        if (out_b) std::memcpy(out_b, h_b, bytes_b);
    }
    return rc;
}

}  // namespace

int kmers::launch_stream_fw(kmers_ctx *ctx, StreamArgs &a, int src_bits, int dst_bits, int n_words, bool vec_ok) {
    return launch_stream<MODE_FW>(ctx, a, src_bits, dst_bits, n_words, vec_ok);
}

extern "C" {

int kmers_fw(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t *out_fw, uint64_t *out_rc,
             int flags, kmers_result *res) {
    if (ctx && !out_fw && seq && kmers_count(seq->n_bases, k, 1)) return fail(ctx, KMERS_E_BADARG, "out_fw is NULL");
    return run_stream(ctx, seq, k, 1, dst_bits, MODE_FW, out_fw, out_rc, false, 0, flags, res);
}

int kmers_canonical(kmers_ctx *ctx, const kmers_seq *seq, int k, int dst_bits, uint64_t *out_kmers,
                    uint64_t *out_hashes, uint64_t seed, int flags, kmers_result *res) {
    return run_stream(ctx, seq, k, 1, dst_bits, MODE_CANON, out_kmers, out_hashes, true, seed, flags, res);
}

int kmers_spaced(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits, uint64_t *out_kmers,
                 int flags, kmers_result *res) {
    if (ctx && !out_kmers && seq && kmers_count(seq->n_bases, k, stride)) return fail(ctx, KMERS_E_BADARG, "out_kmers is NULL");
    return run_stream(ctx, seq, k, stride, dst_bits, MODE_FW, out_kmers, nullptr, false, 0, flags, res);
}

}  // extern "C"
