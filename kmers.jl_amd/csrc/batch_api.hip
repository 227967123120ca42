// batch_api.hip -- batches of records (include/kmers_hip.h): kmers_batch / kmers_batch_spaced / kmers_minhash_batch, what the
// per-record iterators would yield for every LongSubSeq view of a pool, in one launch sequence (ragged_kernels.hpp,
// record_sketch_kernel.hpp).
#include "../../include/kmers_hip.h"

#include "api_common.hpp"
#include "ragged_kernels.hpp"
#include "record_sketch_kernel.hpp"
#include "scan_kernels.hpp"

using namespace kmers;

// The pool as a DST-bit symbol stream: the pool itself when the kmer alphabet has the source's width (Copyable,
// nothing can fail), else the output of the recode pass (stream in stage 4, flag bits in stage 5).
struct PoolStream {
    const uint64_t *stream = nullptr, *flags = nullptr, *any_flag = nullptr;
};
// `launch` false: only make room and say where the stream will be; pool_stream_range then recodes it piece by piece.
static int pool_stream_range(kmers_ctx *ctx, const kmers_seq *pool, const uint64_t *src0, int dst_bits, const PoolStream &ps, uint64_t w_first,
                             uint64_t n_words);
static int pool_stream(kmers_ctx *ctx, const kmers_seq *pool, const uint64_t *src0, uint64_t origin, uint64_t n_src_words, int dst_bits,
                       PoolStream *out, bool launch = true) {
    (void)origin;
    const int sb = pool->src_bits;
    if (sb == dst_bits) {
        out->stream = src0;
        return KMERS_OK;
    }
    const size_t stream_bytes = (size_t)n_src_words * 8 * dst_bits / sb + 16, flag_bytes = (size_t)n_src_words * 8 / sb + 16;
    if (int rc = ensure_stage(ctx, 4, stream_bytes)) return rc;
    RecodeArgs r{};
    r.src = src0;
    r.n_words = n_src_words;
    r.stream = static_cast<uint64_t *>(ctx->stage[4]);
    r.ascii_table = ascii_table(ctx, dst_bits, pool->alphabet);
    if (sb != 2) {
        if (int rc = ensure_stage(ctx, 5, flag_bytes + 16)) return rc;
        r.flags = static_cast<uint64_t *>(ctx->stage[5]);
        r.any_flag = reinterpret_cast<uint64_t *>(static_cast<char *>(ctx->stage[5]) + ((flag_bytes + 7) & ~(size_t)7));
        HIP_TRY(ctx, hipMemsetAsync(r.any_flag, 0, 8, ctx->stream));
    }
    out->stream = r.stream;
    out->flags = r.flags;
    out->any_flag = r.any_flag;
    if (launch) return pool_stream_range(ctx, pool, src0, dst_bits, *out, 0, n_src_words);
    return KMERS_OK;
}

static int pool_stream_range(kmers_ctx *ctx, const kmers_seq *pool, const uint64_t *src0, int dst_bits, const PoolStream &ps, uint64_t w_first,
                             uint64_t n_words) {
    const int sb = pool->src_bits;
    if (sb == dst_bits || n_words == 0) return KMERS_OK;
    RecodeArgs r{};
    r.src = src0;
    r.n_words = n_words;
    r.w_first = w_first;
    r.stream = const_cast<uint64_t *>(ps.stream);
    r.flags = const_cast<uint64_t *>(ps.flags);
    r.any_flag = const_cast<uint64_t *>(ps.any_flag);
    r.ascii_table = ascii_table(ctx, dst_bits, pool->alphabet);
    dim3 rgrid((unsigned)std::min<uint64_t>((n_words + 255) / 256, (uint64_t)ctx->n_cus * 16)), rblock(256);
    if (sb == 4) hipLaunchKernelGGL((recode_kernel<4, 2>), rgrid, rblock, 0, ctx->stream, r);
    else if (sb == 2) hipLaunchKernelGGL((recode_kernel<2, 4>), rgrid, rblock, 0, ctx->stream, r);
    else if (dst_bits == 2) hipLaunchKernelGGL((recode_kernel<8, 2>), rgrid, rblock, 0, ctx->stream, r);
    else hipLaunchKernelGGL((recode_kernel<8, 4>), rgrid, rblock, 0, ctx->stream, r);
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

// EncodeError of a batch: the window that starts at symbol j (0-based) of record r holds the record's first symbol
// that the kmer alphabet cannot encode (all earlier windows of the record were clean); find it on the host.
static int report_window_error(kmers_ctx *ctx, const kmers_seq *pool, const uint64_t *src0, uint64_t origin, const RaggedSpan *d_spans,
                               uint64_t r, uint64_t j, int k, int dst_bits, kmers_result *res) {
    const int sb = pool->src_bits;
    kmers_span bad_span;
    HIP_TRY(ctx, hipMemcpy(&bad_span, d_spans + r, sizeof bad_span, hipMemcpyDeviceToHost));
    const uint64_t p0 = bad_span.first_base + j + origin;     // symbol index from src0
    const uint64_t wlo = p0 * sb / 64, whi = ((p0 + k) * sb + 63) / 64;
    std::vector<uint64_t> w(whi - wlo);
    HIP_TRY(ctx, hipMemcpyAsync(w.data(), src0 + wlo, w.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint8_t table[256];
    if (sb == 8) {
        const uint32_t tb = ascii_table(ctx, dst_bits, pool->alphabet);
        for (uint32_t b = 0; b < 256u; ++b) table[b] = ascii_entry(tb, b);
    }
    for (uint64_t t = 0; t < (uint64_t)k; ++t) {
        const uint64_t bit = (p0 + t) * sb - wlo * 64;
        const uint32_t enc = (uint32_t)((w[bit >> 6] >> (bit & 63u)) & ((1ull << sb) - 1ull));
        const bool bad = sb == 8 ? table[enc] == 0x80 : (sb == 4 && dst_bits == 2 && __builtin_popcount(enc) != 1);
        if (bad) {
            if (res) {
                res->status = KMERS_E_ENCODE;
                res->err_pos = j + t + 1;
                res->err_enc = enc;
                res->n_out = r;
            }
            ctx->last_error = "EncodeError: symbol cannot be encoded in the kmer alphabet";
            return KMERS_E_ENCODE;
        }
    }
    return fail(ctx, KMERS_E_HIP, "kmers_batch: a flagged window holds no offending symbol");
}

static int batch_impl(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int mode, int k,
                      int dst_bits, uint64_t *out_a, uint64_t *out_b, uint64_t seed, uint64_t *out_offsets,
                      uint64_t capacity, int flags, kmers_result *res, uint64_t stride = 1) {
    clear(res);
    if (stride == 0 || stride >= 0xFFFFFFFFull) return fail(ctx, KMERS_E_BADARG, "J must be at least 1 (and below 2^32)");
    if (stride != 1 && mode != KMERS_BATCH_FW) return fail(ctx, KMERS_E_BADARG, "strided batches yield forward kmers (SpacedKmers)");
    if (int rc = check_common(ctx, pool, k, 1, dst_bits, flags & ~(KMERS_ASYNC | KMERS_SPANS_DEVICE | KMERS_BATCH_SKIP | INTERNAL_OUT_DEVICE))) {
        if (res) res->status = rc;
        return rc;
    }
    if (flags & (KMERS_ASYNC | KMERS_OUT_TUPLES)) return fail(ctx, KMERS_E_BADARG, "kmers_batch is synchronous and writes separate arrays");
    if (mode != KMERS_BATCH_FW && mode != KMERS_BATCH_CANONICAL) return fail(ctx, KMERS_E_BADARG, "unknown batch mode");
    if (n_spans && !spans) return fail(ctx, KMERS_E_BADARG, "spans is NULL");
    const int nw = kmers_words_per_kmer(k, dst_bits);
    if (n_spans >= 0xFFFFFFFFull) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_batch supports fewer than 2^32 records per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- the ragged layout, on the device: spans -> HBM, elements per record, exclusive scan
    const uint64_t n = n_spans;
    if (n == 0) {
        if (out_offsets) out_offsets[0] = 0;
        return KMERS_OK;
    }
    const uint64_t n_seg = (n + (uint64_t)LAYOUT_CHUNK * LAYOUT_CHUNKS - 1) / ((uint64_t)LAYOUT_CHUNK * LAYOUT_CHUNKS);  // segments of the layout pass
    const size_t span_bytes = (size_t)n * 16, off_bytes = ((size_t)n + 1) * 8;
    const bool spans_dev = (flags & KMERS_SPANS_DEVICE) != 0;
    if (int rc = ensure_stage(ctx, 3, span_bytes + off_bytes)) return rc;
    char *meta = static_cast<char *>(ctx->stage[3]);
    const RaggedSpan *d_spans = spans_dev ? reinterpret_cast<const RaggedSpan *>(spans) : reinterpret_cast<const RaggedSpan *>(meta);
    uint64_t *d_off = reinterpret_cast<uint64_t *>(meta + span_bytes);
    if (!spans_dev) HIP_TRY(ctx, hipMemcpyAsync(meta, spans, span_bytes, hipMemcpyHostToDevice, ctx->stream));
    // one kernel, nothing cleared before it (scan_kernels.hpp): descriptors tagged with the call's epoch, a ticket counter that only grows
    if (ctx->layout_segs < n_seg) {
        if (ctx->d_layout) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            (void)hipFree(ctx->d_layout);
            ctx->d_layout = nullptr;
            ctx->layout_segs = 0;
        }
        const size_t segs = std::max<uint64_t>(n_seg + n_seg / 2, 4096);
        if (dev_malloc(ctx, reinterpret_cast<void **>(&ctx->d_layout), (4 + 2 * segs) * 8) != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc(batch layout)");
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_layout, 0, (4 + 2 * segs) * 8, ctx->stream));
        ctx->layout_segs = segs;
        ctx->layout_tickets = 0;
        // (the epoch goes on counting: the BAD / ABORT words of d_scratch hold the epoch of an earlier failing call, and a counter
        // that started over would meet it again -- a valid batch reported as a bad span.  The fresh buffer is zeroed, any epoch > 0 is new to it.)
    }
    if (++ctx->layout_epoch >= LAYOUT_EPOCH_LIMIT) {  // (a billion calls on: the tags start over)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_layout + 4, 0, 2 * ctx->layout_segs * 8, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch + LAYOUT_WORD_BAD, 0, 16, ctx->stream));
        ctx->layout_epoch = 1;
    }
    const uint64_t epoch = ctx->layout_epoch;
    {
        LayoutArgs la{};
        la.spans = d_spans;
        la.n = n;
        la.pool_bases = pool->n_bases;
        la.desc = ctx->d_layout + 4;
        la.ticket = ctx->d_layout;
        la.ticket_base = ctx->layout_tickets;
        la.offsets = d_off;
        la.header = reinterpret_cast<unsigned long long *>(ctx->d_scratch);
        la.epoch = (uint32_t)epoch;
        la.k = (uint32_t)k;
        la.step = (uint32_t)stride;
        hipLaunchKernelGGL(ragged_layout_kernel<LAYOUT_CHUNKS>, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, la);
        HIP_TRY(ctx, hipGetLastError());
        ctx->layout_tickets += n_seg;
    }
    uint64_t *const h = ctx->h_result;  // pinned; words 2..5 = [tiles left over, element count, bad span, look-back gave up]
    auto layout_verdict = [&]() -> int {
        if (h[LAYOUT_WORD_ABORT] == epoch) return fail(ctx, KMERS_E_HIP, "kmers_batch: the layout pass gave up waiting for a segment");
        if (h[LAYOUT_WORD_BAD] == epoch) return fail(ctx, KMERS_E_BADARG, "a span reaches outside the pool (or holds 2^32 symbols or more)");
        return KMERS_OK;
    };
    if (out_offsets) HIP_TRY(ctx, hipMemcpyAsync(out_offsets, d_off, off_bytes, hipMemcpyDeviceToHost, ctx->stream));
    // The element count decides the tile layout, and waiting for it costs the call a host round trip in the middle (the device
    // idles between the scan and the element kernel).  A caller that passes device outputs and a sane capacity -- the count it
    // learned from a size query, or an upper bound -- gets the layout sized for the CAPACITY instead, the kernels read the count on
    // the device, and the call waits once, at its end (round 5: the optimistic launch of one-word kmers from a 4-bit pool only).
    // One-word 2-bit kmers out of a 4-bit pool (LongDNA{4}, what BioSequences reads FASTA into) or out of DNA / RNA text (the bytes
    // of a FASTA / FASTQ buffer): the OPTIMISTIC launch below.
    const bool text_pool = pool->src_bits == 8 && pool->alphabet != KMERS_ALPHABET_SYMBOLS;
    const bool opt_shape = (pool->src_bits == 4 || text_pool) && dst_bits == 2 && nw == 1 && stride == 1 && ctx->batch_dense >= 0;
    const bool out_dev_early = (flags & KMERS_MEM_DEVICE) || (flags & INTERNAL_OUT_DEVICE);
    const bool deferred = out_dev_early && (out_a || out_b) && (!out_a || aligned16(out_a)) && (!out_b || aligned16(out_b)) && opt_shape && capacity > 0 &&
                          capacity <= 2 * pool->n_bases + n;
    uint64_t total = capacity;  // (deferred: what the layout is sized for; the real count arrives with the final wait)
    if (!deferred) {
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_scratch, 48, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (int rc = layout_verdict()) return rc;
        total = h[LAYOUT_WORD_TOTAL];
        if (res) res->n_out = total;
        if (total > capacity || (!out_a && !out_b)) {
            if (total > capacity && (out_a || out_b)) {
                if (res) res->status = KMERS_E_CAPACITY;
                return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
            }
            return KMERS_OK;  // size query
        }
        if (total == 0) return KMERS_OK;
    }
    // tile = 1..8 passes of 1024 elements: long tiles amortise the two rounds of loads every tile starts with
    // (each about 5 us under the store load), short ones keep a small batch spread over the device
    uint64_t passes = total / ((uint64_t)RG_UNIT * (uint64_t)ctx->n_cus * 32u);
    passes = std::min<uint64_t>(std::max<uint64_t>(passes, 1), (uint64_t)RG_MAX_PASSES);
    // ... but not longer than the record slots staged in LDS allow (very short reads: many records per pass)
    const uint64_t per_pass = (n * (uint64_t)RG_UNIT + total - 1) / total;  // records per 1024 elements, on average
    passes = std::min<uint64_t>(passes, std::max<uint64_t>(1, (uint64_t)(RG_SLOTS * 7 / 8) / std::max<uint64_t>(per_pass, 1)));
    // ... nor than the stretch of the stream a tile can stage (records lying far apart in the pool: a FASTQ buffer)
    const uint64_t words_per_pass = (uint64_t)(1.1 * (double)pool->n_bases / (double)total * RG_UNIT * dst_bits / 64.0) + 1;
    passes = std::min<uint64_t>(passes, std::max<uint64_t>(1, (uint64_t)(RG_STAGE * 7 / 8) / words_per_pass));
    if (ctx->batch_passes > 0) passes = std::min<uint64_t>((uint64_t)ctx->batch_passes, (uint64_t)RG_MAX_PASSES);  // tests, tuning
    // (the run path of the element kernel takes RG_PASS elements per workgroup pass: whole passes)
    const uint32_t tile_elems = (uint32_t)((passes * RG_UNIT + RG_PASS - 1) / RG_PASS * RG_PASS);
    const uint64_t n_tiles = (total + tile_elems - 1) / tile_elems;
    const bool wide = nw > 4;  // kmers of more than four words: ragged_wide_kernel, one lane per element, no tiles
    if (int rc = ensure_stage(ctx, 6, wide ? 16 : (size_t)n_tiles * sizeof(RaggedTile) + (size_t)n_tiles + 16)) return rc;
    RaggedTile *d_tiles = static_cast<RaggedTile *>(ctx->stage[6]);
    uint8_t *d_status = reinterpret_cast<uint8_t *>(d_tiles + n_tiles);  // one byte per tile: the optimistic launch below

    Staged st;
    if (int rc = stage_sequence(ctx, pool, flags, &st)) return rc;
    const bool dev = flags & KMERS_MEM_DEVICE;
    const int sb = pool->src_bits;
    const uint64_t *src0 = st.d_words + (st.first_bit >> 6);        // word that holds pool symbol 0
    const uint64_t origin = (st.first_bit & 63u) / (uint64_t)sb;     // its symbol offset inside that word
    const uint64_t n_src_words = ((origin + pool->n_bases) * (uint64_t)sb + 63) / 64;

    RaggedArgs a{};
    a.rec_off = d_off;
    a.spans = d_spans;
    a.tiles = d_tiles;
    a.n_records = n;
    a.n_elems = total;
    a.seed = seed;
    a.err_slot = ctx->d_err;
    a.k = (uint32_t)k;
    a.skip = (flags & KMERS_BATCH_SKIP) ? 1u : 0u;
    a.tile = tile_elems;
    a.stride = (uint32_t)stride;
    a.dense = (int32_t)ctx->batch_dense;
    a.stream_origin = origin;
    // The OPTIMISTIC launch first -- no recode pass, every tile recodes its own stretch and takes the dense path
    // (ragged_kernels.hpp), the pool is read once.  Tiles that could not (short or scattered records) count themselves; only then do
    // the recode pass and the general launch run, for those tiles.  Round 6: a symbol the kmer alphabet cannot encode no longer
    // sends its tile there -- the tile makes the flag bits of its own stretch and marks (KMERS_BATCH_SKIP) or reports the elements
    // whose windows hold one; and text is recoded in the tile like a 4-bit pool is (8 bytes per v_perm / v_dot4 pair).
    const bool optimistic = !wide && opt_shape && tile_elems <= (uint32_t)(RG_MAX_PASSES * RG_UNIT);
    unsigned long long *d_redo = reinterpret_cast<unsigned long long *>(ctx->d_scratch + 2);
    if (!wide)  // (the optimistic launch's status bytes and redo count are cleared by this kernel on its way)
        hipLaunchKernelGGL(ragged_tiles_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, d_off, d_spans, n, n_tiles,
                           total, deferred ? d_off + n : nullptr, tile_elems, (uint32_t)k, (uint32_t)stride, (uint32_t)dst_bits, origin, d_tiles,
                           optimistic ? d_status : nullptr, optimistic ? d_redo : nullptr,
                           reinterpret_cast<const unsigned long long *>(ctx->d_scratch + LAYOUT_WORD_ABORT), epoch, capacity);
    HIP_TRY(ctx, hipGetLastError());
    PoolStream ps;
    if (!optimistic) {
        if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps)) return rc;
        a.stream = ps.stream;
        a.flags = ps.flags;
        a.any_flag = ps.any_flag;
    } else {
        a.src_opt = src0;
        a.n_src_words = n_src_words;
        a.opt_from = sb == 8 ? 8u : 4u;
        a.text = sb == 8 ? (pool->alphabet == KMERS_ALPHABET_RNA ? 2u : 1u) : 0u;
        a.tile_status = d_status;
        a.redo_count = d_redo;
    }

    uint64_t *d_a = out_a, *d_b = out_b;
    const bool b_is_hash = mode == KMERS_BATCH_CANONICAL;
    const size_t bytes_a = (size_t)total * nw * 8, bytes_b = (size_t)total * (b_is_hash ? 1 : nw) * 8;
    const bool out_dev = dev || (flags & INTERNAL_OUT_DEVICE);
    if (!out_dev) {
        if (out_a) { if (int rc = ensure_stage(ctx, 1, bytes_a)) return rc; d_a = (uint64_t *)ctx->stage[1]; }
        if (out_b) { if (int rc = ensure_stage(ctx, 2, bytes_b)) return rc; d_b = (uint64_t *)ctx->stage[2]; }
    }
    if ((nw == 2 || nw == 4) && ((d_a && !aligned16(d_a)) || (d_b && !b_is_hash && !aligned16(d_b))))
        return fail(ctx, KMERS_E_BADARG, "two- and four-word kmer outputs must be 16-byte aligned");
    a.out_a = d_a;
    a.out_b = d_b;
    const bool vec = (!d_a || aligned16(d_a)) && (!d_b || aligned16(d_b));
    bool have_result = false;  // the error slot has already travelled to the host (the optimistic launch)
    dim3 grid((unsigned)n_tiles), block(256);
#define RG(DB, NN, MD)                                                                                         \
    do {                                                                                                       \
        if (vec && NN == 1) hipLaunchKernelGGL((ragged_kernel<DB, NN, MD, NN == 1>), grid, block, 0, ctx->stream, a);  \
        else hipLaunchKernelGGL((ragged_kernel<DB, NN, MD, false>), grid, block, 0, ctx->stream, a);          \
    } while (0)
#define RGM(DB, NN) do { if (mode == KMERS_BATCH_FW) RG(DB, NN, MODE_FW); else RG(DB, NN, MODE_CANON); } while (0)
#define RGN(DB) do { if (nw == 1) RGM(DB, 1); else if (nw == 2) RGM(DB, 2); else if (nw == 3) RGM(DB, 3); else RGM(DB, 4); } while (0)
    if (wide) {
        grid = dim3((unsigned)((total + 255) / 256));
        const uint32_t nwu = (uint32_t)nw;
        if (dst_bits == 2) {
            if (mode == KMERS_BATCH_FW) hipLaunchKernelGGL((ragged_wide_kernel<2, MODE_FW>), grid, block, 0, ctx->stream, a, nwu);
            else hipLaunchKernelGGL((ragged_wide_kernel<2, MODE_CANON>), grid, block, 0, ctx->stream, a, nwu);
        } else {
            if (mode == KMERS_BATCH_FW) hipLaunchKernelGGL((ragged_wide_kernel<4, MODE_FW>), grid, block, 0, ctx->stream, a, nwu);
            else hipLaunchKernelGGL((ragged_wide_kernel<4, MODE_CANON>), grid, block, 0, ctx->stream, a, nwu);
        }
    } else if (optimistic && !vec) {  // unaligned outputs: no dense path to be optimistic about
        if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps)) return rc;
        a.stream = ps.stream;
        a.flags = ps.flags;
        a.any_flag = ps.any_flag;
        a.src_opt = nullptr;
        a.tile_status = nullptr;
        RGN(2);
    } else if (optimistic) {
        RGM(2, 1);
        HIP_TRY(ctx, hipGetLastError());
        // (scratch words 0..5 in one copy: with device-resident outputs this is the call's only wait when no tile is left)
        HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_scratch, 48, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (deferred) {  // the count and the span check have arrived with everything else
            if (int rc = layout_verdict()) return rc;
            total = h[LAYOUT_WORD_TOTAL];
            if (res) res->n_out = total;
            if (total > capacity) {
                if (res) res->status = KMERS_E_CAPACITY;
                return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
            }
            if (total == 0) return KMERS_OK;
            a.n_elems = total;
        }
        have_result = out_dev && ctx->h_result[2] == 0;
        if (ctx->h_result[2]) {  // some tiles are left: the recode pass and the general launch for them
            if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps)) return rc;
            a.stream = ps.stream;
            a.flags = ps.flags;
            a.any_flag = ps.any_flag;
            a.src_opt = nullptr;
            RGM(2, 1);
        }
    } else if (dst_bits == 2) RGN(2);
    else RGN(4);
#undef RGN
#undef RGM
#undef RG
    HIP_TRY(ctx, hipGetLastError());
    if (!out_dev) {
        if (out_a) HIP_TRY(ctx, hipMemcpyAsync(out_a, d_a, bytes_a, hipMemcpyDeviceToHost, ctx->stream));
        if (out_b) HIP_TRY(ctx, hipMemcpyAsync(out_b, d_b, bytes_b, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (!have_result) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_scratch, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    const uint64_t g = ctx->h_result[1];
    if (g == NO_ERROR_POS) {
        if (res) res->status = KMERS_OK;
        return KMERS_OK;
    }
    // EncodeError: element g is the first one whose window holds a symbol the kmer alphabet cannot encode;
    // all earlier windows of its record were clean, so its first bad symbol is the record's first
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream));
    std::vector<uint64_t> offs(n + 1);  // rare path: find the record on the host
    HIP_TRY(ctx, hipMemcpyAsync(offs.data(), d_off, off_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t r = (uint64_t)(std::upper_bound(offs.begin(), offs.begin() + n, g) - offs.begin()) - 1;  // last record with off <= g
    const uint64_t j = (g - offs[r]) * stride;                // 0-based start of the window inside the record
    return report_window_error(ctx, pool, src0, origin, d_spans, r, j, k, dst_bits, res);
}

extern "C" {

int kmers_batch_spaced(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k, uint64_t stride,
                       int dst_bits, uint64_t *out_kmers, uint64_t *out_offsets, uint64_t capacity, int flags, kmers_result *res) {
    try {
        return batch_impl(ctx, pool, spans, n_spans, KMERS_BATCH_FW, k, dst_bits, out_kmers, nullptr, 0, out_offsets, capacity, flags, res,
                          stride);
    } catch (const std::bad_alloc &) {
        return fail(ctx, KMERS_E_NOMEM, "host allocation failed in kmers_batch_spaced");
    } catch (...) {
        return fail(ctx, KMERS_E_HIP, "unexpected exception in kmers_batch_spaced");
    }
}

int kmers_batch(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int mode, int k,
                int dst_bits, uint64_t *out_a, uint64_t *out_b, uint64_t seed, uint64_t *out_offsets,
                uint64_t capacity, int flags, kmers_result *res) {
    try {
        return batch_impl(ctx, pool, spans, n_spans, mode, k, dst_bits, out_a, out_b, seed, out_offsets, capacity, flags, res);
    } catch (const std::bad_alloc &) {
        return fail(ctx, KMERS_E_NOMEM, "host allocation failed in kmers_batch");
    } catch (...) {
        return fail(ctx, KMERS_E_HIP, "unexpected exception in kmers_batch");
    }
}

// kmers_minhash_batch, fused: the recode pass (if the pool needs one), then one workgroup per record that derives the
// record's hashes tile by tile and keeps its bottom-s (record_sketch_kernel.hpp).  No layout pass, no hash array.
static int minhash_batch_fused(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k, int dst_bits,
                               uint64_t seed, uint64_t s, uint64_t *out_hashes, uint64_t *out_counts, int flags, kmers_result *res) {
    if (int rc = check_common(ctx, pool, k, 1, dst_bits, flags & ~(KMERS_ASYNC | KMERS_SPANS_DEVICE | KMERS_BATCH_SKIP))) {
        if (res) res->status = rc;
        return rc;
    }
    if (flags & (KMERS_ASYNC | KMERS_OUT_TUPLES)) return fail(ctx, KMERS_E_BADARG, "kmers_minhash_batch is synchronous");
    if (n_spans && !spans) return fail(ctx, KMERS_E_BADARG, "spans is NULL");
    const int nw = kmers_words_per_kmer(k, dst_bits);
    if (n_spans == 0) return KMERS_OK;
    if (n_spans >= 0xFFFFFFFFull) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_minhash_batch supports fewer than 2^32 records per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint64_t n = n_spans;
    const bool spans_dev = (flags & KMERS_SPANS_DEVICE) != 0;
    const size_t span_bytes = (size_t)n * 16;
    if (int rc = ensure_stage(ctx, 3, span_bytes + 16)) return rc;
    char *meta = static_cast<char *>(ctx->stage[3]);
    const RaggedSpan *d_spans = spans_dev ? reinterpret_cast<const RaggedSpan *>(spans) : reinterpret_cast<const RaggedSpan *>(meta);
    uint64_t *d_bad = reinterpret_cast<uint64_t *>(meta + span_bytes);
    if (!spans_dev) HIP_TRY(ctx, hipMemcpyAsync(meta, spans, span_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));

    const bool dev = flags & KMERS_MEM_DEVICE;
    const int sb = pool->src_bits;
    // A large pool in HOST memory whose records lie in pool order (the records of a FASTA file, docs/src/minhash.md:31-41) comes up in
    // PIECES: the copy of piece c + 1 runs beside the recode pass and the sketch kernel of piece c, whose sketches go down before
    // piece c + 2 -- the call takes about as long as the copy of its pool alone (bench.py, the e2e legs: 0.69 -> 0.9 of a plain copy).
    // Anything else (device memory, records out of order or outside the pool, a small pool) takes the one-piece path.
    constexpr uint64_t PIECE_BYTES = 32ull << 20;
    std::vector<uint64_t> cut;  // record indices at which a piece begins, then n
    if (!dev && !spans_dev && (uint64_t)pool->n_bases * sb / 8 >= 4 * PIECE_BYTES && n <= (1ull << 20)) {
        uint64_t prev = 0, begun = 0;
        bool ordered = true;
        cut.push_back(0);
        for (uint64_t i = 0; i < n && ordered; ++i) {
            const uint64_t f = spans[i].first_base, l = spans[i].n_bases;
            ordered = f >= prev && f <= pool->n_bases && l <= pool->n_bases - f;
            prev = f;
            if (i && (f - begun) * sb / 8 >= PIECE_BYTES) {
                cut.push_back(i);
                begun = f;
            } else if (!i) {
                begun = f;
            }
        }
        cut.push_back(n);
        if (!ordered || cut.size() < 4) cut.clear();
    }
    Staged st;
    HostSlice host;
    if (int rc = stage_sequence(ctx, pool, flags, &st, cut.empty() ? nullptr : &host)) return rc;
    const uint64_t *src0 = st.d_words + (st.first_bit >> 6);        // word that holds pool symbol 0
    const uint64_t origin = (st.first_bit & 63u) / (uint64_t)sb;     // its symbol offset inside that word
    const uint64_t n_src_words = ((origin + pool->n_bases) * (uint64_t)sb + 63) / 64;
    PoolStream ps;
    if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps, cut.empty())) return rc;

    const size_t out_bytes = (size_t)n * s * 8, cnt_out_bytes = (size_t)n * 8;
    uint64_t *d_out = out_hashes, *d_cnt = out_counts;
    if (!dev) {
        if (int rc = ensure_stage(ctx, 1, out_bytes)) return rc;
        if (int rc = ensure_stage(ctx, 2, cnt_out_bytes)) return rc;
        d_out = static_cast<uint64_t *>(ctx->stage[1]);
        d_cnt = static_cast<uint64_t *>(ctx->stage[2]);
    }
    // LDS per workgroup = the candidate buffer (+ 0.8 KiB of staged stream).  Short records (one tile of windows) need room
    // for the sketch and that tile: 16 KiB keeps eight workgroups on a CU.  Long records leave about 1.8 s candidates
    // below the provisional threshold: 32 KiB holds them without a merge half way.
    const bool long_records = pool->n_bases / n > 1024;
    const uint32_t run = long_records ? 8u : 4u, rs_tile = 256u * run;
    uint32_t cap = long_records ? 4096u : 2048u;
    while (cap < (uint32_t)s + rs_tile) cap <<= 1;
    if (ctx->sketch_batch_lds == 2048 || ctx->sketch_batch_lds == 4096 || ctx->sketch_batch_lds == 8192)  // tuning (a power of two)
        cap = std::max<uint32_t>(cap, (uint32_t)ctx->sketch_batch_lds);
    RecordSketchArgs a{};
    a.stream = ps.stream;
    a.flags = ps.flags;
    a.any_flag = ps.any_flag;
    a.stream_origin = origin;
    a.spans = d_spans;
    a.pool_bases = pool->n_bases;
    a.seed = seed;
    a.out = d_out;
    a.counts = d_cnt;
    a.err_slot = ctx->d_err;
    a.bad = d_bad;
    a.k = (uint32_t)k;
    a.s = (uint32_t)s;
    a.skip = (flags & KMERS_BATCH_SKIP) ? 1u : 0u;
    a.cap = cap;
    a.n_words = (uint32_t)nw;
    const size_t lds = ((size_t)cap + RS_STAGE + RS_FSTAGE) * 8;
    dim3 grid((unsigned)n), block(256);
#define RS(DB, NN, RR)                                                                                                                  \
    do {                                                                                                                                \
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(record_sketch_kernel<DB, NN, RR>),                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SEG_VALUES + RS_STAGE + RS_FSTAGE) * 8))); \
        hipLaunchKernelGGL((record_sketch_kernel<DB, NN, RR>), grid, block, lds, ctx->stream, a);                                       \
    } while (0)
#define RSR(DB, NN) do { if (run == 8u) RS(DB, NN, 8); else RS(DB, NN, 4); } while (0)
#define RSN(DB) do { if (nw == 1) RSR(DB, 1); else if (nw == 2) RSR(DB, 2); else if (nw == 3) RSR(DB, 3); else if (nw == 4) RSR(DB, 4); else RSR(DB, 0); } while (0)
    ctx->last_batch_pieces = cut.empty() ? 1 : cut.size() - 1;
    if (cut.empty()) {
        if (dst_bits == 2) RSN(2);
        else RSN(4);
        HIP_TRY(ctx, hipGetLastError());
        if (!dev) {
            HIP_TRY(ctx, hipMemcpyAsync(out_hashes, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(out_counts, d_cnt, cnt_out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        }
    } else {
        if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (auto &e : ctx->pipe_events)
            if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const size_t pieces = cut.size() - 1;
        // stage byte b is host byte b; piece c needs the words of its records: [word of its first symbol, word behind the furthest end)
        auto word_of = [&](uint64_t sym) { return ((origin + sym) * (uint64_t)sb) >> 6; };
        std::vector<uint64_t> w_lo(pieces), w_hi(pieces);
        for (size_t c = 0; c < pieces; ++c) {
            uint64_t end = 0;
            for (uint64_t i = cut[c]; i < cut[c + 1]; ++i) end = std::max<uint64_t>(end, spans[i].first_base + spans[i].n_bases);
            w_lo[c] = word_of(spans[cut[c]].first_base) & ~(uint64_t)63;  // (whole 64-word groups: the recode pass writes 1-4 bytes per word)
            w_hi[c] = std::min<uint64_t>(n_src_words, word_of(end) + 2);
            if (c) w_hi[c] = std::max(w_hi[c], w_hi[c - 1]);
        }
        HIP_TRY(ctx, hipEventRecord(ctx->pipe_events[0], ctx->stream));  // (the spans, the flag resets: in front of the first copy)
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->pipe_events[0], 0));
        uint64_t up = w_lo[0];  // words [w_lo[0], up) are on the device
        auto copy_up = [&](size_t c) -> int {
            if (w_hi[c] > up) {
                const size_t b0 = (size_t)up * 8, b1 = std::min<size_t>((size_t)w_hi[c] * 8, host.bytes);
                if (b1 > b0)
                    HIP_TRY(ctx, hipMemcpyAsync(static_cast<char *>(ctx->stage[0]) + b0, host.from + b0, b1 - b0, hipMemcpyHostToDevice, ctx->copy_stream));
                up = w_hi[c];
            }
            HIP_TRY(ctx, hipEventRecord(ctx->pipe_events[1 + (c & 1)], ctx->copy_stream));
            return KMERS_OK;
        };
        if (int rc = copy_up(0)) return rc;
        for (size_t c = 0; c < pieces; ++c) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->pipe_events[1 + (c & 1)], 0));
            if (int rc = pool_stream_range(ctx, pool, src0, dst_bits, ps, w_lo[c], w_hi[c] - w_lo[c])) return rc;
            grid = dim3((unsigned)(cut[c + 1] - cut[c]));
            a.rec_base = cut[c];
            if (dst_bits == 2) RSN(2);
            else RSN(4);
            HIP_TRY(ctx, hipGetLastError());
            // the next piece's copy is enqueued BEFORE this piece's sketches go down: a copy from or to pageable memory holds the
            // host until it is done, and the device should have the next piece's kernels to run by then
            if (c + 1 < pieces)
                if (int rc = copy_up(c + 1)) return rc;
            const size_t r0 = (size_t)cut[c], r1 = (size_t)cut[c + 1];
            HIP_TRY(ctx, hipMemcpyAsync(out_hashes + r0 * s, d_out + r0 * s, (r1 - r0) * s * 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_TRY(ctx, hipMemcpyAsync(out_counts, d_cnt, cnt_out_bytes, hipMemcpyDeviceToHost, ctx->stream));  // (one copy: every copy to pageable memory is a wait)
    }
#undef RSR
#undef RSN
#undef RS
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_scratch, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result + 2, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t g = ctx->h_result[1];
    if (g != NO_ERROR_POS) HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream));
    if (ctx->h_result[2]) return fail(ctx, KMERS_E_BADARG, "a span reaches outside the pool (or holds 2^32 symbols or more)");
    if (g != NO_ERROR_POS)  // (record << 32 | window): the first failing record in batch order, its first failing window
        return report_window_error(ctx, pool, src0, origin, d_spans, g >> 32, g & 0xFFFFFFFFull, k, dst_bits, res);
    if (res) {
        res->status = KMERS_OK;
        res->n_out = n_spans;
    }
    return KMERS_OK;
}

static int minhash_batch_impl(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k, int dst_bits,
                              uint64_t seed, uint64_t s, uint64_t *out_hashes, uint64_t *out_counts, int flags, kmers_result *res) {
    clear(res);
    if (!ctx) return KMERS_E_BADARG;
    if (s == 0 || s > SEG_VALUES / 4) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_minhash_batch supports sketch sizes 1..2048");
    if (n_spans && (!out_hashes || !out_counts)) return fail(ctx, KMERS_E_BADARG, "out_hashes / out_counts is NULL");
    return minhash_batch_fused(ctx, pool, spans, n_spans, k, dst_bits, seed, s, out_hashes, out_counts, flags, res);
}

int kmers_minhash_batch(kmers_ctx *ctx, const kmers_seq *pool, const kmers_span *spans, uint64_t n_spans, int k, int dst_bits,
                        uint64_t seed, uint64_t s, uint64_t *out_hashes, uint64_t *out_counts, int flags, kmers_result *res) {
    try {
        return minhash_batch_impl(ctx, pool, spans, n_spans, k, dst_bits, seed, s, out_hashes, out_counts, flags, res);
    } catch (const std::bad_alloc &) {
        return fail(ctx, KMERS_E_NOMEM, "host allocation failed in kmers_minhash_batch");
    } catch (...) {
        return fail(ctx, KMERS_E_HIP, "unexpected exception in kmers_minhash_batch");
    }
}

int kmers_last_batch_pieces(kmers_ctx *ctx, uint64_t *pieces) {
    if (!ctx || !pieces) return KMERS_E_BADARG;
    *pieces = ctx->last_batch_pieces;
    return KMERS_OK;
}

}  // extern "C"

#ifdef KMERS_RG_PROBE
extern "C" int kmers_debug_rg_probe(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(kmers::rg_probe), 128) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(kmers::rg_probe), z, 128) != hipSuccess) return -1;
    }
    return 0;
}
#endif
