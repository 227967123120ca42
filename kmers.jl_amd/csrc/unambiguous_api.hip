// unambiguous_api.hip -- kmers_unambiguous (include/kmers_hip.h): UnambiguousKmers collected in ONE pass over the source
// (src/iterators/UnambiguousKmers.jl:59-148; unambiguous_kernel.hpp), and the XOR mode of the same kernel for the fused reducer.
#include "../../include/kmers_hip.h"

#include "api_common.hpp"
#include "unambiguous_kernel.hpp"

using namespace kmers;

namespace {

static_assert(DESC_VALUE == DESCRIPTOR_COUNT_MASK, "kmers_sync decodes a tile descriptor with context.hpp's mask");

// UnambiguousKmers over a sequence in which every window survives (a 2-bit source, or a count pass that kept
// everything): its elements are FwKmers plus the start indices 1, 2, ..., so the stream kernel writes them at
// its two-array rate -- no compaction, no offsets.
int emit_all_kept(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, uint64_t n, uint64_t *d_k, long long *d_s) {
    StreamArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_bases = seq->n_bases;
    a.n_kmers = n;
    a.inspect_end = seq->n_bases;
    a.out_a = d_k;
    a.out_starts = d_s;
    a.start_origin = seq->index_origin;
    a.err_slot = ctx->d_err;
    a.err_origin = seq->index_origin;
    a.k = (uint32_t)k;
    a.stride = 1;
    a.ascii_table = ascii_table(ctx, 2, seq->alphabet);
    const bool vec_ok = (!d_k || aligned16(d_k)) && (!d_s || aligned16(d_s));
    return launch_stream_fw(ctx, a, seq->src_bits, 2, kmers_words_per_kmer(k, 2), vec_ok);
}

// Launch of the single-pass UnambiguousKmers kernel (unambiguous_kernel.hpp) in one of its modes.
template <int UMODE>
void launch_unambiguous(kmers_ctx *ctx, int src_bits, int nw, dim3 grid, const UnambArgs &a) {
    dim3 block(BLOCK);
#define UW(SB)                                                                                                  \
    do {                                                                                                        \
        switch (nw) {                                                                                           \
            case 1: hipLaunchKernelGGL((unambiguous_kernel<SB, 1, UMODE>), grid, block, 0, ctx->stream, a); break; \
            case 2: hipLaunchKernelGGL((unambiguous_kernel<SB, 2, UMODE>), grid, block, 0, ctx->stream, a); break; \
            case 3: hipLaunchKernelGGL((unambiguous_kernel<SB, 3, UMODE>), grid, block, 0, ctx->stream, a); break; \
            case 4: hipLaunchKernelGGL((unambiguous_kernel<SB, 4, UMODE>), grid, block, 0, ctx->stream, a); break; \
            default: hipLaunchKernelGGL((unambiguous_kernel<SB, 0, UMODE>), grid, block, 0, ctx->stream, a); break; /* run-time width */ \
        }                                                                                                       \
    } while (0)
    if constexpr (UMODE == UMODE_COUNT) {  // counting does not depend on the kmer width: one instantiation per source
        (void)nw;
        if (src_bits == 8) hipLaunchKernelGGL((unambiguous_kernel<8, 1, UMODE_COUNT>), grid, block, 0, ctx->stream, a);
        else if (src_bits == 4) hipLaunchKernelGGL((unambiguous_kernel<4, 1, UMODE_COUNT>), grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL((unambiguous_kernel<2, 1, UMODE_COUNT>), grid, block, 0, ctx->stream, a);
    } else {
        if (src_bits == 8) UW(8);
        else if (src_bits == 4) UW(4);
        else UW(2);
    }
#undef UW
}

// longest kmer the single-pass kernel stages (a tile and its K-1 symbols of overlap must fit the LDS stream)
constexpr int UNAMB_MAX_K = 30720;
static_assert(UNAMB_MAX_K <= (int)UTILE_MAX - 2048, "a tile and its K - 1 symbols of overlap must fit the LDS stream");
uint32_t unambiguous_tile(kmers_ctx *ctx, int k, int stride) {
    // candidate starts per tile: a multiple of 1024, at most UTILE_MAX.  Long tiles keep the rate of
    // tile descriptors low enough for the look-back; very long kmers leave room for their overlap.
    uint32_t t = ctx->tile_kmers > 0 ? (uint32_t)std::min<int64_t>(ctx->tile_kmers, UTILE_MAX) : UTILE_MAX;
    if (k > 128) t = std::min<uint32_t>(t, (UTILE_MAX - (uint32_t)k) / UROUND * UROUND);
    t = std::max<uint32_t>(UROUND, t / UROUND * UROUND);
    // a stride lattice: a tile length that is a multiple of the stride too puts the lattice at the same place in every tile (the
    // kernel then needs no division per tile), if at least half of the length survives
    if (stride > 1) {
        uint64_t g = UROUND, s = (uint64_t)stride;
        while (s) { const uint64_t r = g % s; g = s; s = r; }      // gcd(1024, stride)
        const uint64_t unit = (uint64_t)UROUND / g * (uint64_t)stride;  // lcm
        if (unit <= t && t / unit * unit >= t / 2) t = (uint32_t)(t / unit * unit);
    }
    return t;
}

// UnambiguousKmers: ONE pass over the source (unambiguous_kernel.hpp): tile descriptors + decoupled look-back inside the
// emitting kernel; the element count is known when the kernel has run.
int run_unambiguous(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, uint64_t *out_kmers,
                    int64_t *out_starts, uint64_t capacity, int flags, kmers_result *res) {
    const int nw = kmers_words_per_kmer(k, 2);
    uint64_t n = kmers_count(seq->n_bases, k, 1);
    const bool ascii = seq->src_bits == 8;
    // An ASCII source is scanned to its end even when it is shorter than K: an invalid byte
    // still throws (UnambiguousKmers.jl:117-123).  Validate with 1-symbol windows, emit nothing.
    const bool validate_only = ascii && n == 0 && seq->n_bases > 0;
    if (n == 0 && !validate_only) return KMERS_OK;
    if (validate_only) {
        n = seq->n_bases;
        k = 1;
    }
    const bool tuples = (flags & KMERS_OUT_TUPLES) != 0;
    if (tuples && out_starts) return fail(ctx, KMERS_E_BADARG, "KMERS_OUT_TUPLES: out_starts must be NULL");
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
    const bool dev = flags & KMERS_MEM_DEVICE;
    const bool query = !out_kmers && !out_starts;  // size query: count only

    UnambArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_cand = n;
    a.n_bases = seq->n_bases;
    // text: the reference's ASCII_SKIPPING_LUT; a collection of symbols: its generic method (UnambiguousKmers.jl:88-106)
    a.ascii_table = seq->alphabet == KMERS_ALPHABET_SYMBOLS ? (uint32_t)SYMBOL_TABLE_SKIPPING : (uint32_t)ASCII_TABLE_SKIPPING;
    a.err_slot = ctx->d_err;
    a.k = (uint32_t)k;
    a.stride = (uint32_t)stride;
    a.index_origin = seq->index_origin;
    a.tile_starts = unambiguous_tile(ctx, k, stride);
    a.tile_phase = stride > 1 && a.tile_starts % (uint32_t)stride != 0 ? 1u : 0u;
    a.n_words = (uint32_t)nw;
    a.n_tiles = (n + a.tile_starts - 1) / a.tile_starts;
    a.tuples = tuples ? 1u : 0u;
    a.stamps = reinterpret_cast<uint64_t *>(ctx->stamps_ptr);
    const uint64_t cap_grid = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (uint64_t)1 << 30;

    // a 2-bit source has no ambiguous symbols: every start survives, nothing to resolve (kmers of more than four
    // words take the run-time-width instantiation of the one-pass kernel like every other source)
    const bool known_all = seq->src_bits == 2 && stride == 1 && !validate_only && nw <= 4;
    uint64_t total = n;
    // Host-memory outputs are staged through HBM buffers of exactly `total` elements, so the host path counts first
    // (it is PCIe-bound anyway); device outputs and their capacity are used as they are: one pass.
    const bool count_first = !known_all && (query || validate_only || !dev);
    if (count_first) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, 8, ctx->stream));
        a.total = reinterpret_cast<unsigned long long *>(ctx->d_scratch);
        dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, std::min<uint64_t>(cap_grid, (uint64_t)ctx->n_cus * 8)));
        launch_unambiguous<UMODE_COUNT>(ctx, seq->src_bits, nw, grid, a);
        HIP_TRY(ctx, hipGetLastError());
        // an invalid byte anywhere in an ASCII source is an EncodeError (collect reads the error slot and the count)
        if (int erc = collect(ctx, res, 0, &total)) return erc;
    }
    if (validate_only) total = 0;
    if (res) res->n_out = total;
    if (query) return KMERS_OK;
    if (known_all || count_first) {
        if (total > capacity) {
            if (res) res->status = KMERS_E_CAPACITY;
            return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
        }
        if (total == 0) return KMERS_OK;
    }
    uint64_t *d_k = out_kmers;
    long long *d_s = reinterpret_cast<long long *>(out_starts);
    const size_t kb = (size_t)total * (tuples ? nw + 1 : nw) * 8, sb = (size_t)total * 8;
    if (!dev) {
        if (out_kmers) { if (int rc = ensure_stage(ctx, 1, kb)) return rc; d_k = (uint64_t *)ctx->stage[1]; }
        if (out_starts) { if (int rc = ensure_stage(ctx, 2, sb)) return rc; d_s = (long long *)ctx->stage[2]; }
    }
    if (known_all && !tuples) {
        // nothing can be dropped: FwKmers + start indices at the stream kernel's rate
        if (int rc = emit_all_kept(ctx, seq, st, k, n, d_k, d_s)) return rc;
    } else {
        if (int rc = ensure_stage(ctx, 3, ((size_t)a.n_tiles + 2) * 8)) return rc;
        unsigned long long *scratch = static_cast<unsigned long long *>(ctx->stage[3]);
        HIP_TRY(ctx, hipMemsetAsync(scratch, 0, ((size_t)a.n_tiles + 2) * 8, ctx->stream));
        a.desc = scratch;
        a.ticket = scratch + a.n_tiles;
        a.abort_flag = scratch + a.n_tiles + 1;
        a.out_kmers = d_k;
        a.out_starts = d_s;
        a.capacity = dev ? capacity : total;
        a.vec16 = ((!d_k || aligned16(d_k)) && (!d_s || aligned16(d_s))) ? 1u : 0u;
        // a persistent grid: every workgroup draws tickets until none is left (UNAMB_EMIT_WGS workgroups per CU: three, 46 KiB of
        // LDS each; unambiguous_kernel.hpp says why not four)
        dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, std::min<uint64_t>(cap_grid, (uint64_t)ctx->n_cus * UNAMB_EMIT_WGS)));
        launch_unambiguous<UMODE_EMIT>(ctx, seq->src_bits, nw, grid, a);
        HIP_TRY(ctx, hipGetLastError());
        if (flags & KMERS_ASYNC) {
            // enqueue only: the count (the last tile's inclusive prefix) follows the kernel into pinned memory and kmers_sync
            // reports it -- together with KMERS_E_CAPACITY if it exceeds `capacity` (nothing was stored beyond it)
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result + 8, a.desc + (a.n_tiles - 1), 24, hipMemcpyDeviceToHost, ctx->stream));
            ctx->unamb_pending = true;
            ctx->unamb_capacity = capacity;
            if (res) res->n_out = 0;
            return KMERS_OK;
        }
        // the last tile's inclusive prefix is the element count
        uint64_t *h = ctx->h_result + 2;  // pinned: [last descriptor, ticket counter, abort flag]
        HIP_TRY(ctx, hipMemcpyAsync(h, a.desc + (a.n_tiles - 1), 24, hipMemcpyDeviceToHost, ctx->stream));
        if (ascii) {  // an invalid byte anywhere in the source is an EncodeError
            if (int erc = collect(ctx, res, 0)) return erc;
        } else {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (h[2]) return fail(ctx, KMERS_E_HIP, "UnambiguousKmers: a tile never published its count (look-back gave up)");
        total = h[0] & DESC_VALUE;
        if (res) res->n_out = total;
        if (total > capacity) {  // the kernel stored nothing at or beyond the capacity
            if (res) res->status = KMERS_E_CAPACITY;
            return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
        }
    }
    if (!dev) {
        if (out_kmers) HIP_TRY(ctx, hipMemcpyAsync(out_kmers, d_k, kb, hipMemcpyDeviceToHost, ctx->stream));
        if (out_starts) HIP_TRY(ctx, hipMemcpyAsync(out_starts, d_s, sb, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (flags & KMERS_ASYNC) return KMERS_OK;  // (a 2-bit source: the count is known, res->n_out holds it; the launch is enqueued)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}

}  // namespace

// UnambiguousKmers under the fused XOR reducer (kmers_reduce_xor_iter): the single-pass kernel's XOR mode (no descriptors, no
// look-back: nothing is placed).  Arguments already checked by the caller.
int kmers::unambiguous_xor(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, uint64_t *out_value, int flags, kmers_result *res) {
    if (k > UNAMB_MAX_K) return fail(ctx, KMERS_E_UNSUPPORTED, "fused UnambiguousKmers reducer: K above 30720");
    uint64_t n = kmers_count(seq->n_bases, k, 1);
    const bool ascii = seq->src_bits == 8;
    const bool validate_only = ascii && n == 0 && seq->n_bases > 0;  // invalid bytes still throw (UnambiguousKmers.jl:117-123)
    if (n == 0 && !validate_only) return KMERS_OK;
    int kk = k;
    if (validate_only) {
        n = seq->n_bases;
        kk = 1;
    }
    Staged st;
    if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, 8, ctx->stream));
    UnambArgs a{};
    a.src = st.d_words;
    a.first_bit = st.first_bit;
    a.n_cand = n;
    a.n_bases = seq->n_bases;
    // text: the reference's ASCII_SKIPPING_LUT; a collection of symbols: its generic method (UnambiguousKmers.jl:88-106)
    a.ascii_table = seq->alphabet == KMERS_ALPHABET_SYMBOLS ? (uint32_t)SYMBOL_TABLE_SKIPPING : (uint32_t)ASCII_TABLE_SKIPPING;
    a.err_slot = ctx->d_err;
    a.k = (uint32_t)kk;
    a.stride = (uint32_t)stride;
    a.index_origin = seq->index_origin;
    a.tile_starts = kk > 128 ? (UTILE_MAX - (uint32_t)kk) / UROUND * UROUND : UTILE_MAX;  // nothing is streamed out: long tiles
    a.tile_phase = stride > 1 && a.tile_starts % (uint32_t)stride != 0 ? 1u : 0u;
    a.n_words = (uint32_t)kmers_words_per_kmer(kk, 2);
    a.n_tiles = (n + a.tile_starts - 1) / a.tile_starts;
    a.total = reinterpret_cast<unsigned long long *>(ctx->d_scratch);
    dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, (uint64_t)ctx->n_cus * 8));
    launch_unambiguous<UMODE_XOR>(ctx, seq->src_bits, kmers_words_per_kmer(kk, 2), grid, a);
    HIP_TRY(ctx, hipGetLastError());
    uint64_t value = 0;
    const int rc = collect(ctx, res, 0, &value);
    if (rc == KMERS_OK && !validate_only) *out_value = value;
    return rc;
}

extern "C" {

int kmers_unambiguous(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, uint64_t *out_kmers,
                      int64_t *out_starts, uint64_t capacity, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, stride, 2, flags)) {
        if (res) res->status = rc;
        return rc;
    }
    if ((flags & KMERS_ASYNC) && !(flags & KMERS_MEM_DEVICE)) return fail(ctx, KMERS_E_BADARG, "KMERS_ASYNC requires KMERS_MEM_DEVICE");
    if ((flags & KMERS_ASYNC) && !out_kmers && !out_starts) return fail(ctx, KMERS_E_BADARG, "the size query of kmers_unambiguous is synchronous");
    if (k > UNAMB_MAX_K) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_unambiguous: K above 30720");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_unambiguous(ctx, seq, k, stride, out_kmers, out_starts, capacity, flags, res);
}

}  // extern "C"
