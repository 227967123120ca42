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

// One launch over the elements [kmer0, kmer0 + nk) of the iterator: the whole call, or one chunk of a host-pointer call.
// `st` holds the staged source (its word 0 contains the view's first symbol at st.first_bit).
int launch_range(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int stride, int dst_bits, int mode, bool tuples, uint64_t seed,
                 uint64_t kmer0, uint64_t nk, uint64_t *d_a, uint64_t *d_b, bool b_is_hash, int flags) {
    const int nw = kmers_words_per_kmer(k, dst_bits);
    const uint64_t n = nk;
    StreamArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit + kmer0 * (uint64_t)stride * (uint64_t)seq->src_bits;
    a.n_bases = (n - 1) * (uint64_t)stride + (uint64_t)k;
    a.n_kmers = n;
    a.inspect_end = (n - 1) * (uint64_t)stride + (uint64_t)k;  // end of the last kmer (== n_bases for stride 1)
    a.out_a = d_a;
    a.out_b = d_b;
    a.seed = seed;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin + kmer0 * (uint64_t)stride;
    a.k = (uint32_t)k;
    a.stride = (uint32_t)stride;
    a.ascii_table = ascii_table(ctx, dst_bits, seq->alphabet);
    a.tuples = tuples ? 1u : 0u;
    (void)b_is_hash;

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
    return rc;
}

// Host-pointer calls whose outputs are large: the elements in CHUNKS through two sets of device buffers -- the kernel of chunk
// c + 1 runs while chunk c travels to the host on a second stream -- instead of one launch into device copies of the whole
// outputs (16 GB of HBM for a Gbase of canonical kmers + hashes) followed by one copy.  The call is bound by the copy to the
// host either way (16.5 bytes per kmer over PCIe against 0.6 ms per Gbase of kernel); what the chunks buy is the kernel time
// and the source's H2D off the critical path, and a device footprint of four chunks.  (collect(CanonicalDNAMers{31}(seq)) of a
// Julia host takes this path: src/iterators/CanonicalKmers.jl:220-225 element by element.)
constexpr size_t PIPE_MIN_BYTES = (size_t)96 << 20;    // outputs smaller than this: one launch, one copy
constexpr size_t PIPE_CHUNK_BYTES = (size_t)32 << 20;  // per output array and chunk
int run_chunked(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int stride, int dst_bits, int mode, bool tuples, uint64_t seed,
                uint64_t n, uint64_t *out_a, uint64_t *out_b, size_t ea, size_t eb, bool b_is_hash, int flags, kmers_result *res) {
    if (!ctx->copy_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &e : ctx->pipe_events) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    const size_t per = std::max(ea, eb);
    const uint64_t CH = std::max<uint64_t>(4096, PIPE_CHUNK_BYTES / per / 4096 * 4096);  // elements per chunk (16-byte aligned outputs)
    if (out_a) { if (int rc = ensure_stage(ctx, 1, 2 * CH * ea)) return rc; }
    if (out_b) { if (int rc = ensure_stage(ctx, 2, 2 * CH * eb)) return rc; }
    char *const dA = static_cast<char *>(ctx->stage[1]), *const dB = static_cast<char *>(ctx->stage[2]);
    hipEvent_t *ev_kernel = ctx->pipe_events, *ev_copied = ctx->pipe_events + 2;
    const uint64_t nc = (n + CH - 1) / CH;
    auto launch = [&](uint64_t c) -> int {
        const int set = (int)(c & 1u);
        const uint64_t k0 = c * CH, nk = std::min<uint64_t>(CH, n - k0);
        if (c >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ev_copied[set], 0));  // the chunk before last has left this set
        if (int rc = launch_range(ctx, seq, st, k, stride, dst_bits, mode, tuples, seed, k0, nk, out_a ? (uint64_t *)(dA + set * CH * ea) : nullptr,
                                  out_b ? (uint64_t *)(dB + set * CH * eb) : nullptr, b_is_hash, flags))
            return rc;
        HIP_TRY(ctx, hipEventRecord(ev_kernel[set], ctx->stream));
        return KMERS_OK;
    };
    if (int rc = launch(0)) return rc;
    for (uint64_t c = 0; c < nc; ++c) {
        if (c + 1 < nc) {
            if (int rc = launch(c + 1)) return rc;  // (before the copy of chunk c is issued: into pageable memory that call blocks)
        }
        const int set = (int)(c & 1u);
        const uint64_t k0 = c * CH, nk = std::min<uint64_t>(CH, n - k0);
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ev_kernel[set], 0));
        if (out_a) HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<char *>(out_a) + k0 * ea, dA + set * CH * ea, nk * ea, hipMemcpyDeviceToHost, ctx->copy_stream));
        if (out_b) HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<char *>(out_b) + k0 * eb, dB + set * CH * eb, nk * eb, hipMemcpyDeviceToHost, ctx->copy_stream));
        HIP_TRY(ctx, hipEventRecord(ev_copied[set], ctx->copy_stream));
    }
    const int rc = collect(ctx, res, n);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    return rc;
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
    const size_t ea = (tuples ? tuple_words : (size_t)nw) * 8, eb = (b_is_hash ? 1 : (size_t)nw) * 8;  // bytes per element
    const size_t bytes_a = (size_t)n * ea, bytes_b = (size_t)n * eb;
    if (!dev && (out_a ? bytes_a : 0) + (out_b ? bytes_b : 0) >= PIPE_MIN_BYTES && ctx->host_chunks >= 0)
        return run_chunked(ctx, seq, st, k, stride, dst_bits, mode, tuples, seed, n, out_a, out_b, ea, eb, b_is_hash, flags, res);
    if (!dev) {
        if (out_a) { if (int rc = ensure_stage(ctx, 1, bytes_a)) return rc; d_a = (uint64_t *)ctx->stage[1]; }
        if (out_b) { if (int rc = ensure_stage(ctx, 2, bytes_b)) return rc; d_b = (uint64_t *)ctx->stage[2]; }
    }
    if ((nw == 2 || nw == 4) && ((d_a && !aligned16(d_a)) || (d_b && !b_is_hash && !aligned16(d_b))))
        return fail(ctx, KMERS_E_BADARG, "two- and four-word kmer outputs must be 16-byte aligned");
    if (tuples && !aligned16(d_a)) return fail(ctx, KMERS_E_BADARG, "tuple outputs must be 16-byte aligned");

    ctx->call_flags = flags;  // (the launcher may time its shape table inside a synchronous call only, stream_launch.hpp)
    int rc = launch_range(ctx, seq, st, k, stride, dst_bits, mode, tuples, seed, 0, n, d_a, d_b, b_is_hash, flags);
    ctx->call_flags = KMERS_ASYNC;
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
