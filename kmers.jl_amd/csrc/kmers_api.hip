// kmers_api.hip -- the C ABI of libkmers_hip.so (include/kmers_hip.h): context, argument
// checking (the reference's constructor errors), host<->HBM staging and kernel launches.
// There is deliberately no CPU compute path here.
#include "../../include/kmers_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ascii_tables.hpp"
#include "context.hpp"
#include "batch_kernels.hpp"
#include "unambiguous_kernel.hpp"
#include "wide_kernel.hpp"
#include "composition_kernel.hpp"
#include "ragged_kernels.hpp"
#include "record_sketch_kernel.hpp"
#include "run_kernel.hpp"
#include "stream_kernel.hpp"

using namespace kmers;

namespace {

// Pageable host memory costs HIP about 10-15 us per small copy (internal staging + waits); a call on a
// short sequence with host pointers makes three of them.  Copies that fit go through pinned memory instead.
constexpr size_t BOUNCE_IN = 256 << 10, BOUNCE_OUT = 1 << 20;
constexpr int INTERNAL_OUT_DEVICE = 1 << 16;  // batch_impl: out_a / out_b are device pointers even if the pool is host memory

void clear(kmers_result *res) {
    if (res) std::memset(res, 0, sizeof *res);
}

// Common argument checks.  K, J errors mirror the constructors (FwKmers.jl:31-35,
// SpacedKmers.jl:26-32); geometry limits are this library's.
int check_common(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits, int flags, bool any_width = false) {
    if (!ctx) return KMERS_E_BADARG;
    if (!seq) return fail(ctx, KMERS_E_BADARG, "seq is NULL");
    if (k < 1) return fail(ctx, KMERS_E_BADARG, "K must be at least 1");
    if (stride < 1) return fail(ctx, KMERS_E_BADARG, "J must be at least 1");
    if (seq->src_bits != 2 && seq->src_bits != 4 && seq->src_bits != 8)
        return fail(ctx, KMERS_E_BADARG, "src_bits must be 2, 4 or 8 (ASCII bytes)");
    if (seq->src_bits == 8 && (seq->alphabet < 0 || seq->alphabet > KMERS_ALPHABET_SYMBOLS))
        return fail(ctx, KMERS_E_BADARG, "alphabet of a byte source must be 0 (DNA text), 1 (RNA text) or 2 (symbol values)");
    if (seq->src_bits == 8 && (flags & KMERS_MEM_DEVICE) && (reinterpret_cast<uintptr_t>(seq->words) & 7u))
        return fail(ctx, KMERS_E_BADARG, "device ASCII buffers must be 8-byte aligned");
    if (seq->n_bases && !seq->words) return fail(ctx, KMERS_E_BADARG, "seq.words is NULL");
    if ((flags & KMERS_ASYNC) && !(flags & KMERS_MEM_DEVICE))
        return fail(ctx, KMERS_E_BADARG, "KMERS_ASYNC requires KMERS_MEM_DEVICE");
    if (dst_bits != 2 && dst_bits != 4) return fail(ctx, KMERS_E_BADARG, "dst_bits must be 2 or 4");
    // the iterators take kmers of any width (wide_kernel.hpp); the other entry points kmers of one to four words
    if (!any_width && n_coding_elements(k, dst_bits) > 4)
        return fail(ctx, KMERS_E_UNSUPPORTED, "this entry point takes kmers of at most four words (K <= 128 two-bit, K <= 64 four-bit)");
    return KMERS_OK;
}

struct Staged {
    const uint64_t *d_words = nullptr;  // device pointer whose word 0 holds first_bit
    uint64_t first_bit = 0;
};

// Make the sequence words available in HBM.  Host memory: copy the words the view touches.
int stage_sequence(kmers_ctx *ctx, const kmers_seq *seq, int flags, Staged *out) {
    uint64_t bit0 = seq->first_base * (uint64_t)seq->src_bits;
    if (flags & KMERS_MEM_DEVICE) {
        out->d_words = seq->words;
        out->first_bit = bit0;
        return KMERS_OK;
    }
    if (seq->src_bits == 8) {  // bytes: copy exactly the view (the host pointer need not be aligned)
        size_t nbytes = (size_t)seq->n_bases;
        if (int rc = ensure_stage(ctx, 0, nbytes + 16)) return rc;
        const void *from = reinterpret_cast<const char *>(seq->words) + seq->first_base;
        if (nbytes && nbytes <= BOUNCE_IN) {
            std::memcpy(ctx->h_bounce, from, nbytes);
            from = ctx->h_bounce;
        }
        if (nbytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], from, nbytes, hipMemcpyHostToDevice, ctx->stream));
        out->d_words = static_cast<const uint64_t *>(ctx->stage[0]);
        out->first_bit = 0;
        return KMERS_OK;
    }
    uint64_t w0 = bit0 >> 6;
    uint64_t w1 = (bit0 + seq->n_bases * (uint64_t)seq->src_bits + 63) >> 6;
    size_t bytes = (size_t)(w1 - w0) * 8;
    if (int rc = ensure_stage(ctx, 0, bytes + 8)) return rc;
    const void *from = seq->words + w0;
    if (bytes && bytes <= BOUNCE_IN) {
        std::memcpy(ctx->h_bounce, from, bytes);
        from = ctx->h_bounce;
    }
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], from, bytes, hipMemcpyHostToDevice, ctx->stream));
    out->d_words = static_cast<const uint64_t *>(ctx->stage[0]);
    out->first_bit = bit0 & 63u;
    return KMERS_OK;
}

// Wait for the stream and turn the device error slot into a kmers_result.
int collect(kmers_ctx *ctx, kmers_result *res, uint64_t n_out, uint64_t *value_out = nullptr) {
    // one 16-byte copy into pinned memory: the reduction word (if any) and the error slot
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_scratch, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned long long pos = ctx->h_result[1];
    if (value_out) *value_out = ctx->h_result[0];
    if (pos == NO_ERROR_POS) {
        if (res) {
            res->status = KMERS_OK;
            res->n_out = n_out;
        }
        return KMERS_OK;
    }
    // EncodeError: the slot holds error_key = (global 0-based position << 8) | raw symbol, written by the kernel at fault
    // time (stream_kernel.hpp); nothing is read back from the sequence, which the caller of an asynchronous launch may
    // already have released.  Re-arm the slot.
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (res) {
        res->status = KMERS_E_ENCODE;
        res->err_pos = (uint64_t)(pos >> 8) + 1;
        res->err_enc = (uint32_t)(pos & 0xffull);
        res->n_out = 0;
    }
    ctx->last_error = "EncodeError: symbol cannot be encoded in the kmer alphabet";
    return KMERS_E_ENCODE;
}

// table id of ascii_entry(): BioSequences.ascii_encode of the kmer alphabet (ascii_tables.hpp)
// (kmers_seq.alphabet: 0 = DNA text, 1 = RNA text, 2 = one BioSymbols value per byte -- GenericRecoding sources)
uint32_t ascii_table(kmers_ctx *, int dst_bits, int alphabet) {
    if (alphabet == KMERS_ALPHABET_SYMBOLS) return dst_bits == 4 ? (uint32_t)SYMBOL_TABLE_4BIT : (uint32_t)SYMBOL_TABLE_2BIT;
    return (dst_bits == 4 ? 2u : 0u) + (alphabet != 0 ? 1u : 0u);
}

// Default tile: about 16 KiB of output per workgroup (four 16-byte stores per lane), one tile
// per workgroup.  Measured on MI355X (profiles/r01_tuning.md): shorter workgroups are bound by
// workgroup launch + the exposed source-load latency, longer ones and persistent grid-stride
// loops lose 10-20 % of the HBM write rate.
static uint32_t default_tile(uint32_t out_bytes_per_kmer, uint32_t pass) {
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
    const uint32_t pass = ((stride1 || pair) && n_words == 1 ? 2u : 1u) * BLOCK;  // kmers per workgroup pass
    uint32_t out_bytes = 8u * n_words * ((a.out_a ? 1u : 0u) + (MODE == MODE_FW && a.out_b ? 1u : 0u)) +
                         (MODE == MODE_CANON && a.out_b ? 8u : 0u) + (MODE == MODE_FW && a.out_starts ? 8u : 0u);
    if (a.tuples) out_bytes = MODE == MODE_FW ? 16u * n_words : 8u * n_words + 8u;
    if (MODE == MODE_XOR || MODE == MODE_SKETCH || MODE == MODE_COUNT) out_bytes = 4u;  // nothing streamed out: long tiles
    if (MODE == MODE_MINIMIZER) out_bytes = 8u * n_words;
    uint32_t max_tile_symbols = (uint32_t)MAX_TILE_BITS / (uint32_t)dst_bits;
    if (MODE == MODE_MINIMIZER) max_tile_symbols -= std::min<uint32_t>(max_tile_symbols / 2, a.window_kmers);  // room for the longer overlap
    uint32_t tile = ctx->tile_kmers > 0 ? (uint32_t)ctx->tile_kmers : default_tile(out_bytes, pass);
    // strided tiles read `stride` times the source per element: twice the output per workgroup amortises the longer load
    // phase (SpacedDNAMers{21,3} over 1 Gbase: 0.64-0.68 -> 0.71-0.74 of 8 TB/s, profiles/r02_tuning.md)
    if (ctx->tile_kmers <= 0 && J > 1 && MODE == MODE_FW) tile *= 2;
    tile = std::min<uint32_t>(tile, max_tile_symbols / J);
    tile = std::max<uint32_t>(pass, tile / pass * pass);
    if ((uint64_t)(tile - 1) * J + 1 > (uint64_t)max_tile_symbols) return fail(ctx, KMERS_E_UNSUPPORTED, "stride too large for the tile kernel");
    a.tile_kmers = tile;
    a.n_tiles = (a.n_kmers + tile - 1) / tile;
    a.stamps = reinterpret_cast<uint64_t *>(ctx->stamps_ptr);
    uint64_t cap = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (uint64_t)1 << 30;
    dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, cap));
    dim3 block(BLOCK);
    if (src_bits == 8 && dst_bits == 2) launch_widths<MODE, 8, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 8) launch_widths<MODE, 8, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 4 && dst_bits == 2) launch_widths<MODE, 4, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 2 && dst_bits == 2) launch_widths<MODE, 2, 2>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else if (src_bits == 4 && dst_bits == 4) launch_widths<MODE, 4, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    else launch_widths<MODE, 2, 4>(n_words, stride1, pair, fwd, grid, block, ctx->stream, a, dyn_lds);
    HIP_TRY(ctx, hipGetLastError());
    return KMERS_OK;
}

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

bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Shared body of kmers_fw / kmers_canonical / kmers_spaced.
int run_stream(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits, int mode,
               uint64_t *out_a, uint64_t *out_b, bool b_is_hash, uint64_t seed, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, stride, dst_bits, flags, true)) {
        if (res) res->status = rc;
        return rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nw = kmers_words_per_kmer(k, dst_bits);
    if (nw > 4 && (flags & KMERS_OUT_TUPLES)) return fail(ctx, KMERS_E_UNSUPPORTED, "KMERS_OUT_TUPLES: kmers of at most four words");
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
    if (nw > 4) {
        // kmers of more than four words: the run-time-width kernel (wide_kernel.hpp), one lane per kmer
        dim3 grid((unsigned)((n + BLOCK - 1) / BLOCK)), block(BLOCK);
        const uint32_t nwu = (uint32_t)nw;
#define WIDE(SB, DB)                                                                                         \
    do {                                                                                                     \
        if (mode == MODE_FW) hipLaunchKernelGGL((wide_kernel<SB, DB, MODE_FW>), grid, block, 0, ctx->stream, a, nwu);   \
        else hipLaunchKernelGGL((wide_kernel<SB, DB, MODE_CANON>), grid, block, 0, ctx->stream, a, nwu);                \
    } while (0)
        if (seq->src_bits == 8 && dst_bits == 2) WIDE(8, 2);
        else if (seq->src_bits == 8) WIDE(8, 4);
        else if (seq->src_bits == 4 && dst_bits == 2) WIDE(4, 2);
        else if (seq->src_bits == 2 && dst_bits == 2) WIDE(2, 2);
        else if (seq->src_bits == 4 && dst_bits == 4) WIDE(4, 4);
        else WIDE(2, 4);
#undef WIDE
        HIP_TRY(ctx, hipGetLastError());
        rc = KMERS_OK;
    } else if ((uint64_t)stride * (uint64_t)dst_bits > 64) {
        // gather path (forward kmers only: kmers_spaced); a tile would stage mostly unused symbols
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
    return launch_stream<MODE_FW>(ctx, a, seq->src_bits, 2, kmers_words_per_kmer(k, 2), vec_ok);
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
constexpr int UNAMB_MAX_K = (int)UTILE_MAX - 2048;
uint32_t unambiguous_tile(kmers_ctx *ctx, int k) {
    // candidate starts per tile: a multiple of 1024 (one wavefront round), at most UTILE_MAX.  Long tiles keep the rate of
    // tile descriptors low enough for the look-back; very long kmers leave room for their overlap.
    uint32_t t = ctx->tile_kmers > 0 ? (uint32_t)std::min<int64_t>(ctx->tile_kmers, UTILE_MAX) : UTILE_MAX;
    if (k > 128) t = std::min<uint32_t>(t, (UTILE_MAX - (uint32_t)k) / UROUND * UROUND);
    return std::max<uint32_t>(UROUND, t / UROUND * UROUND);
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
    a.tile_starts = unambiguous_tile(ctx, k);
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
        // a persistent grid: every workgroup draws tickets until none is left (UNAMB_EMIT_WGS = six workgroups per CU, 24.2 KiB
        // of LDS and 80 VGPRs each; the kernel's time falls with every resident workgroup, profiles/r02_tuning.md section 6)
        dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, std::min<uint64_t>(cap_grid, (uint64_t)ctx->n_cus * UNAMB_EMIT_WGS)));
        launch_unambiguous<UMODE_EMIT>(ctx, seq->src_bits, nw, grid, a);
        HIP_TRY(ctx, hipGetLastError());
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
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}

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

// Fused consumers of one- and two-word 2-bit kmers (K <= 64): the rolling run kernel (run_kernel.hpp);
// everything else (three- and four-word kmers, 4-bit kmer alphabets) goes through the stream kernel's fused modes.
template <int RMODE, int SMODE>
int launch_consumer(kmers_ctx *ctx, const kmers_seq *seq, const Staged &st, int k, int dst_bits, StreamArgs &a, size_t best_bytes = 0) {
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
    const uint64_t resident = ctx->max_grid > 0 ? (uint64_t)ctx->max_grid : (uint64_t)ctx->n_cus * 8;  // persistent grid
    dim3 grid((unsigned)std::min<uint64_t>(a.n_tiles, resident)), block(RBLOCK);
#define RUNK(SB)                                                                                             \
    do {                                                                                                     \
        if (RMODE == RMODE_XOR && !a.xor_canonical) {                                                        \
            if (k <= 32) hipLaunchKernelGGL((run_kernel<SB, RMODE_XOR, 1, false>), grid, block, best_bytes, ctx->stream, a); \
            else hipLaunchKernelGGL((run_kernel<SB, RMODE_XOR, 2, false>), grid, block, best_bytes, ctx->stream, a);         \
        } else if (k <= 32) hipLaunchKernelGGL((run_kernel<SB, RMODE, 1, true>), grid, block, best_bytes, ctx->stream, a);   \
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

int kmers_abi_version(void) { return KMERS_ABI_VERSION; }

int kmers_words_per_kmer(int k, int dst_bits) {
    if (k < 0 || (dst_bits != 2 && dst_bits != 4 && dst_bits != 8)) return -1;
    return n_coding_elements(k, dst_bits);
}

uint64_t kmers_count(uint64_t n_bases, int k, int stride) {
    if (k < 1 || stride < 1 || n_bases < (uint64_t)k) return 0;
    return (n_bases - (uint64_t)k) / (uint64_t)stride + 1;  // SpacedKmers.jl:41; stride 1 == FwKmers.jl:42
}

int kmers_shard_plan(uint64_t n_bases, int k, uint64_t stride, int src_bits, int n_shards, int shard_id,
                     kmers_shard *out) {
    if (!out || k < 1 || stride < 1 || n_shards < 1 || shard_id < 0 || shard_id >= n_shards) return KMERS_E_BADARG;
    if (src_bits != 2 && src_bits != 4 && src_bits != 8) return KMERS_E_BADARG;
    const uint64_t bits = (uint64_t)src_bits, per_word = 64 / bits, K = (uint64_t)k, n = (uint64_t)n_shards;
    const uint64_t m = n_bases < K ? 0 : (n_bases - K) / stride + 1;
    const uint64_t total_words = (n_bases * bits + 63) / 64;
    // shard boundaries sit on source-word boundaries AND on the stride lattice
    uint64_t a = stride, b = per_word;
    while (b) { uint64_t t = a % b; a = b; b = t; }
    const uint64_t unit_kmers = per_word / a;  // lcm(stride, per_word) / stride
    uint64_t per = (m + n - 1) / n;
    per = per ? (per + unit_kmers - 1) / unit_kmers * unit_kmers : unit_kmers;
    const uint64_t words_per_shard = per / unit_kmers * (stride / a);  // per * stride / per_word
    const uint64_t overhang = K > stride ? K - stride : 0;             // symbols a shard's last window reaches past it
    const uint64_t halo = (overhang * bits + 63) / 64;
    const uint64_t g = (uint64_t)shard_id;
    auto plan = [&](uint64_t q, kmers_shard *o) {
        const uint64_t lo = m < q * per ? m : q * per, hi = m < (q + 1) * per ? m : (q + 1) * per;
        const uint64_t fw = total_words < q * words_per_shard ? total_words : q * words_per_shard;
        uint64_t lw = total_words;
        if (q != n - 1 && (q + 1) * words_per_shard < total_words) lw = (q + 1) * words_per_shard;
        const uint64_t nk = hi - lo;
        const uint64_t need_end = nk ? (((lo + nk - 1) * stride + K) * bits + 63) / 64 : fw;
        uint64_t h = 0;
        if (q < n - 1 && need_end > lw) h = need_end - lw < halo ? need_end - lw : halo;
        o->first_kmer = lo;
        o->n_kmers = nk;
        o->first_base = lo * stride;
        o->n_bases = nk ? (nk - 1) * stride + K : 0;
        o->first_word = fw;
        o->n_own_words = lw - fw;
        o->halo_words = (uint32_t)h;
        o->send_words = 0;
    };
    if (m == 0 || (n > 1 && words_per_shard < halo + 1)) {  // too short to shard: shard 0 does it all
        *out = kmers_shard{g ? m : 0, g ? 0 : m, g ? m * stride : 0, g ? 0 : n_bases, g ? total_words : 0,
                           g ? 0 : total_words, 0, 0};
        return KMERS_OK;
    }
    plan(g, out);
    if (g > 0) {
        kmers_shard left;
        plan(g - 1, &left);
        out->send_words = left.halo_words;
    }
    return KMERS_OK;
}

int kmers_supported(int src_bits, int dst_bits, int k, int stride) {
    if (src_bits != 2 && src_bits != 4 && src_bits != 8) return 0;
    if (dst_bits != 2 && dst_bits != 4) return 0;
    if (k < 1 || stride < 1) return 0;
    return 1;  // FwKmers / FwRvIterator / CanonicalKmers / SpacedKmers take kmers of any width (wide_kernel.hpp)
}

int kmers_ctx_create(int device, void *hip_stream, kmers_ctx **out) {
    if (!out) return KMERS_E_BADARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return KMERS_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return KMERS_E_HIP;
    kmers_ctx *ctx = new (std::nothrow) kmers_ctx();
    if (!ctx) return KMERS_E_NOMEM;
    ctx->device = device;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return KMERS_E_HIP; }
        ctx->own_stream = true;
    }
    if (hipMalloc(&ctx->d_scratch, 64 * 8) != hipSuccess || hipHostMalloc(&ctx->h_result, 64, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&ctx->h_bounce, BOUNCE_IN + BOUNCE_OUT, hipHostMallocDefault) != hipSuccess ||
        (ctx->d_err = reinterpret_cast<unsigned long long *>(ctx->d_scratch + 1)) == nullptr ||
        hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        kmers_ctx_destroy(ctx);
        return KMERS_E_HIP;
    }
    *out = ctx;
    return KMERS_OK;
}

void kmers_ctx_destroy(kmers_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->stage)
        if (p) (void)hipFree(p);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_result) (void)hipHostFree(ctx->h_result);
    if (ctx->h_bounce) (void)hipHostFree(ctx->h_bounce);
    if (ctx->d_recent) (void)hipFree(ctx->d_recent);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *kmers_ctx_stream(kmers_ctx *ctx) { return ctx ? ctx->stream : nullptr; }
const char *kmers_last_error(kmers_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "no context"; }

int kmers_ctx_set_param(kmers_ctx *ctx, int param, int64_t value) {
    if (!ctx) return KMERS_E_BADARG;
    if (param == KMERS_PARAM_TILE_KMERS) ctx->tile_kmers = value;
    else if (param == KMERS_PARAM_MAX_GRID) ctx->max_grid = value;
    else if (param == KMERS_PARAM_STAMPS_PTR) ctx->stamps_ptr = value;
    else if (param == KMERS_PARAM_SKETCH_HOST_ONLY) ctx->sketch_host_only = value != 0;
    else if (param == KMERS_PARAM_BATCH_PASSES) ctx->batch_passes = value;
    else if (param == KMERS_PARAM_SKETCH_BATCH_LDS) ctx->sketch_batch_lds = value;
    else return fail(ctx, KMERS_E_BADARG, "unknown parameter");
    return KMERS_OK;
}

int kmers_sync(kmers_ctx *ctx, kmers_result *res) {
    if (!ctx) return KMERS_E_BADARG;
    clear(res);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return collect(ctx, res, 0);
}

int kmers_dev_alloc(kmers_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc", e);
    return KMERS_OK;
}
int kmers_dev_free(kmers_ctx *ctx, void *p) {
    if (!ctx) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(p));
    return KMERS_OK;
}
int kmers_memcpy_h2d(kmers_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}
int kmers_memcpy_d2h(kmers_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}

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
        if ((uint64_t)stride * (uint64_t)dst_bits > 64) return fail(ctx, KMERS_E_UNSUPPORTED, "fused SpacedKmers reducer: stride * bits per symbol must be <= 64");
        const uint64_t n = kmers_count(seq->n_bases, k, stride);
        if (n == 0) return KMERS_OK;
        Staged st;
        if (int rc = stage_sequence(ctx, seq, flags, &st)) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_scratch, 0, 8, ctx->stream));
        StreamArgs a{};
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
    // UnambiguousKmers: the single-pass kernel's XOR mode (no descriptors, no look-back: nothing is placed)
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
        hipError_t e = hipMalloc(&ctx->d_recent, (size_t)RECENT_SLOTS * 8);
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
    if ((uint64_t)stride * (uint64_t)dst_bits > 64 * 8) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_minimizers supports window strides up to 512 / dst_bits symbols");
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
    // strides >= span leave gaps the reference's loop never reads: restrict the validation to the windows
    if (int rc = launch_stream<MODE_MINIMIZER>(ctx, a, seq->src_bits, dst_bits, nw, false)) return rc;
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
    if (k > 12) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_composition supports K <= 12 (4^K counters)");
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
            // 4^11 and 4^12 counters: 64+ passes would cost more than memory-side global atomics (37 ms per Gbase)
            StreamArgs a{};
            a.out_a = reinterpret_cast<uint64_t *>(d_counts);
            if (int rc = launch_fused<MODE_COUNT>(ctx, seq, st, k, 2, a, 0)) return rc;
        }
    }
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(out_counts, d_counts, bins * 4, hipMemcpyDeviceToHost, ctx->stream));
    return collect(ctx, res, n);
}

// The pool as a DST-bit symbol stream: the pool itself when the kmer alphabet has the source's width (Copyable,
// nothing can fail), else the output of the recode pass (stream in stage 4, flag bits in stage 5).
struct PoolStream {
    const uint64_t *stream = nullptr, *flags = nullptr, *any_flag = nullptr;
};
static int pool_stream(kmers_ctx *ctx, const kmers_seq *pool, const uint64_t *src0, uint64_t origin, uint64_t n_src_words, int dst_bits,
                       PoolStream *out) {
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
    if (n_src_words) {
        dim3 rgrid((unsigned)std::min<uint64_t>((n_src_words + 255) / 256, (uint64_t)ctx->n_cus * 16)), rblock(256);
        if (sb == 4) hipLaunchKernelGGL((recode_kernel<4, 2>), rgrid, rblock, 0, ctx->stream, r);
        else if (sb == 2) hipLaunchKernelGGL((recode_kernel<2, 4>), rgrid, rblock, 0, ctx->stream, r);
        else if (dst_bits == 2) hipLaunchKernelGGL((recode_kernel<8, 2>), rgrid, rblock, 0, ctx->stream, r);
        else hipLaunchKernelGGL((recode_kernel<8, 4>), rgrid, rblock, 0, ctx->stream, r);
        HIP_TRY(ctx, hipGetLastError());
    }
    out->stream = r.stream;
    out->flags = r.flags;
    out->any_flag = r.any_flag;
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
    if (nw > 4) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_batch supports kmers of at most four words");
    if (n_spans >= 0xFFFFFFFFull) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_batch supports fewer than 2^32 records per call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // ---- the ragged layout, on the device: spans -> HBM, elements per record, exclusive scan
    const uint64_t n = n_spans;
    if (n == 0) {
        if (out_offsets) out_offsets[0] = 0;
        return KMERS_OK;
    }
    const uint64_t n_seg = (n + SCAN_SEG - 1) / SCAN_SEG;
    const size_t span_bytes = (size_t)n * 16, cnt_bytes = ((size_t)n * 4 + 15) & ~(size_t)15, off_bytes = ((size_t)n + 1) * 8,
                 seg_bytes = ((size_t)n_seg + 2) * 8;
    const bool spans_dev = (flags & KMERS_SPANS_DEVICE) != 0;
    if (int rc = ensure_stage(ctx, 3, span_bytes + cnt_bytes + off_bytes + seg_bytes)) return rc;
    char *meta = static_cast<char *>(ctx->stage[3]);
    const RaggedSpan *d_spans = spans_dev ? reinterpret_cast<const RaggedSpan *>(spans) : reinterpret_cast<const RaggedSpan *>(meta);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(meta + span_bytes);
    uint64_t *d_off = reinterpret_cast<uint64_t *>(meta + span_bytes + cnt_bytes);
    uint64_t *d_seg = reinterpret_cast<uint64_t *>(meta + span_bytes + cnt_bytes + off_bytes);  // [n_seg + 1], then the bad-span flag
    uint64_t *d_bad = d_seg + n_seg + 1;
    if (!spans_dev) HIP_TRY(ctx, hipMemcpyAsync(meta, spans, span_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(d_bad, 0, 8, ctx->stream));
    {
        dim3 g((unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)ctx->n_cus * 16)), b(256);
        hipLaunchKernelGGL(ragged_count_kernel, g, b, 0, ctx->stream, d_spans, n, (uint32_t)k, (uint32_t)stride, pool->n_bases, d_cnt, d_bad);
        hipLaunchKernelGGL(scan_segment_sums_kernel, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, d_cnt, n, d_seg);
        hipLaunchKernelGGL(scan_segments_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_seg, n_seg);
        hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)n_seg), dim3(256), 0, ctx->stream, d_cnt, n, d_seg, n_seg, d_off);
        HIP_TRY(ctx, hipGetLastError());
    }
    uint64_t *h = reinterpret_cast<uint64_t *>(ctx->h_bounce);  // pinned: [total, bad]
    HIP_TRY(ctx, hipMemcpyAsync(h, d_off + n, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h + 1, d_bad, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (out_offsets) HIP_TRY(ctx, hipMemcpyAsync(out_offsets, d_off, off_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t total = h[0];
    if (h[1]) return fail(ctx, KMERS_E_BADARG, "a span reaches outside the pool (or holds 2^32 symbols or more)");
    if (res) res->n_out = total;
    if (total > capacity || (!out_a && !out_b)) {
        if (total > capacity && (out_a || out_b)) {
            if (res) res->status = KMERS_E_CAPACITY;
            return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
        }
        return KMERS_OK;  // size query
    }
    if (total == 0) return KMERS_OK;
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
    if (int rc = ensure_stage(ctx, 6, (size_t)n_tiles * sizeof(RaggedTile))) return rc;
    RaggedTile *d_tiles = static_cast<RaggedTile *>(ctx->stage[6]);

    Staged st;
    if (int rc = stage_sequence(ctx, pool, flags, &st)) return rc;
    const bool dev = flags & KMERS_MEM_DEVICE;
    const int sb = pool->src_bits;
    const uint64_t *src0 = st.d_words + (st.first_bit >> 6);        // word that holds pool symbol 0
    const uint64_t origin = (st.first_bit & 63u) / (uint64_t)sb;     // its symbol offset inside that word
    const uint64_t n_src_words = ((origin + pool->n_bases) * (uint64_t)sb + 63) / 64;
    hipLaunchKernelGGL(ragged_tiles_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, ctx->stream, d_off, d_spans, n, n_tiles,
                       total, tile_elems, (uint32_t)k, (uint32_t)stride, (uint32_t)dst_bits, origin, d_tiles);
    HIP_TRY(ctx, hipGetLastError());

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
    a.stream_origin = origin;
    PoolStream ps;
    if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps)) return rc;
    a.stream = ps.stream;
    a.flags = ps.flags;
    a.any_flag = ps.any_flag;

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
    dim3 grid((unsigned)n_tiles), block(256);
#define RG(DB, NN, MD)                                                                                         \
    do {                                                                                                       \
        if (vec && NN == 1) hipLaunchKernelGGL((ragged_kernel<DB, NN, MD, NN == 1>), grid, block, 0, ctx->stream, a);  \
        else hipLaunchKernelGGL((ragged_kernel<DB, NN, MD, false>), grid, block, 0, ctx->stream, a);          \
    } while (0)
#define RGM(DB, NN) do { if (mode == KMERS_BATCH_FW) RG(DB, NN, MODE_FW); else RG(DB, NN, MODE_CANON); } while (0)
#define RGN(DB) do { if (nw == 1) RGM(DB, 1); else if (nw == 2) RGM(DB, 2); else if (nw == 3) RGM(DB, 3); else RGM(DB, 4); } while (0)
    if (dst_bits == 2) RGN(2);
    else RGN(4);
#undef RGN
#undef RGM
#undef RG
    HIP_TRY(ctx, hipGetLastError());
    if (!out_dev) {
        if (out_a) HIP_TRY(ctx, hipMemcpyAsync(out_a, d_a, bytes_a, hipMemcpyDeviceToHost, ctx->stream));
        if (out_b) HIP_TRY(ctx, hipMemcpyAsync(out_b, d_b, bytes_b, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_result, ctx->d_scratch, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
    if (nw > 4) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_minhash_batch supports kmers of at most four words");
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

    Staged st;
    if (int rc = stage_sequence(ctx, pool, flags, &st)) return rc;
    const bool dev = flags & KMERS_MEM_DEVICE;
    const int sb = pool->src_bits;
    const uint64_t *src0 = st.d_words + (st.first_bit >> 6);        // word that holds pool symbol 0
    const uint64_t origin = (st.first_bit & 63u) / (uint64_t)sb;     // its symbol offset inside that word
    const uint64_t n_src_words = ((origin + pool->n_bases) * (uint64_t)sb + 63) / 64;
    PoolStream ps;
    if (int rc = pool_stream(ctx, pool, src0, origin, n_src_words, dst_bits, &ps)) return rc;

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
    const size_t lds = ((size_t)cap + RS_STAGE + RS_FSTAGE) * 8;
    dim3 grid((unsigned)n), block(256);
#define RS(DB, NN, RR)                                                                                                                  \
    do {                                                                                                                                \
        HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(record_sketch_kernel<DB, NN, RR>),                              \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)((SEG_VALUES + RS_STAGE + RS_FSTAGE) * 8))); \
        hipLaunchKernelGGL((record_sketch_kernel<DB, NN, RR>), grid, block, lds, ctx->stream, a);                                       \
    } while (0)
#define RSR(DB, NN) do { if (run == 8u) RS(DB, NN, 8); else RS(DB, NN, 4); } while (0)
#define RSN(DB) do { if (nw == 1) RSR(DB, 1); else if (nw == 2) RSR(DB, 2); else if (nw == 3) RSR(DB, 3); else RSR(DB, 4); } while (0)
    if (dst_bits == 2) RSN(2);
    else RSN(4);
#undef RSR
#undef RSN
#undef RS
    HIP_TRY(ctx, hipGetLastError());
    if (!dev) {
        HIP_TRY(ctx, hipMemcpyAsync(out_hashes, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(out_counts, d_cnt, cnt_out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
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

int kmers_unambiguous(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, uint64_t *out_kmers,
                      int64_t *out_starts, uint64_t capacity, int flags, kmers_result *res) {
    clear(res);
    if (int rc = check_common(ctx, seq, k, stride, 2, flags & ~KMERS_ASYNC, true)) {
        if (res) res->status = rc;
        return rc;
    }
    if (flags & KMERS_ASYNC) return fail(ctx, KMERS_E_BADARG, "kmers_unambiguous is synchronous (data-dependent count)");
    if (k > UNAMB_MAX_K) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_unambiguous: K above 30720");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return run_unambiguous(ctx, seq, k, stride, out_kmers, out_starts, capacity, flags, res);
}

int kmers_fx_hash(kmers_ctx *ctx, const uint64_t *kmers, int n_words, uint64_t n, uint64_t seed, uint64_t *out,
                  int flags) {
    if (!ctx) return KMERS_E_BADARG;
    if (n_words < 0 || (n && (!kmers || !out) && n_words > 0) || (n && !out)) return fail(ctx, KMERS_E_BADARG, "bad fx_hash arguments");
    if (n == 0) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool dev = flags & KMERS_MEM_DEVICE;
    const uint64_t *d_in = kmers;
    uint64_t *d_out = out;
    size_t in_bytes = (size_t)n * n_words * 8, out_bytes = (size_t)n * 8;
    if (!dev) {
        if (int rc = ensure_stage(ctx, 0, in_bytes + 8)) return rc;
        if (int rc = ensure_stage(ctx, 1, out_bytes)) return rc;
        if (in_bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], kmers, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_in = (const uint64_t *)ctx->stage[0];
        d_out = (uint64_t *)ctx->stage[1];
    }
    // one pass per workgroup (short-lived workgroups write fastest); NW == 1 takes two kmers per lane
    // and needs 16-byte aligned arrays, else it falls back to the generic-width kernel
    const bool pair_ok = n_words == 1 && aligned16(d_in) && aligned16(d_out);
    const uint64_t work_items = pair_ok ? (n + 1) / 2 : n;
    dim3 block(256), grid((unsigned)std::min<uint64_t>((work_items + 255) / 256, (uint64_t)1 << 30));
    if (n_words == 1 && !pair_ok) n_words = -1;
    switch (n_words) {
        case 0: hipLaunchKernelGGL(fx_hash_kernel_any, grid, block, 0, ctx->stream, d_in, 0, n, seed, d_out); break;  // 0-mer: the seed
        case 1: hipLaunchKernelGGL((fx_hash_kernel<1>), grid, block, 0, ctx->stream, d_in, n, seed, d_out); break;
        case 2: hipLaunchKernelGGL((fx_hash_kernel<2>), grid, block, 0, ctx->stream, d_in, n, seed, d_out); break;
        case -1: hipLaunchKernelGGL(fx_hash_kernel_any, grid, block, 0, ctx->stream, d_in, 1, n, seed, d_out); break;
        default: hipLaunchKernelGGL(fx_hash_kernel_any, grid, block, 0, ctx->stream, d_in, n_words, n, seed, d_out); break;
    }
    HIP_TRY(ctx, hipGetLastError());
    if (flags & KMERS_ASYNC) return KMERS_OK;
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}

int kmers_transform(kmers_ctx *ctx, int op, const uint64_t *kmers, int k, int bits, uint64_t n, uint64_t *out,
                    int flags) {
    if (!ctx) return KMERS_E_BADARG;
    if (op < 0 || op > 8 || k < 1 || (bits != 2 && bits != 4)) return fail(ctx, KMERS_E_BADARG, "bad transform arguments");
    if (op == KMERS_OP_COUNT_GC && bits != 2) return fail(ctx, KMERS_E_UNSUPPORTED, "count(isGC) is defined for 2-bit kmers (src/counting.jl:1)");
    const int nw = n_coding_elements(k, bits);
    if (nw > 4) return fail(ctx, KMERS_E_UNSUPPORTED, "kmers_transform supports up to 4 words per kmer");
    if ((op == KMERS_OP_AS_INTEGER || op == KMERS_OP_FROM_INTEGER) && nw > 2)
        return fail(ctx, KMERS_E_BADARG, "Must have at most 128 bits in encoding (src/kmer.jl:324)");
    if (n == 0) return KMERS_OK;
    if (!kmers || !out) return fail(ctx, KMERS_E_BADARG, "NULL kmer array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool dev = flags & KMERS_MEM_DEVICE;
    const uint64_t *d_in = kmers;
    uint64_t *d_out = out;
    size_t in_bytes = (size_t)n * nw * 8, out_bytes = (size_t)n * ((op == KMERS_OP_ISCANONICAL || op == KMERS_OP_COUNT_GC) ? 1 : nw) * 8;
    if (!dev) {
        if (int rc = ensure_stage(ctx, 0, in_bytes + 8)) return rc;
        if (int rc = ensure_stage(ctx, 1, out_bytes)) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], kmers, in_bytes, hipMemcpyHostToDevice, ctx->stream));
        d_in = (const uint64_t *)ctx->stage[0];
        d_out = (uint64_t *)ctx->stage[1];
    }
    const bool vec = aligned16(d_in) && aligned16(d_out) && nw != 3;
    const uint64_t items = (vec && nw == 1) ? (n + 1) / 2 : n;
    dim3 block(256), grid((unsigned)std::min<uint64_t>((items + 255) / 256, (uint64_t)1 << 30));  // one pass per workgroup
#define TL(NW_, B_)                                                                                                      \
    do {                                                                                                                 \
        if (vec) hipLaunchKernelGGL((transform_kernel<NW_, B_, true>), grid, block, 0, ctx->stream, op, d_in, n, k, d_out);  \
        else hipLaunchKernelGGL((transform_kernel<NW_, B_, false>), grid, block, 0, ctx->stream, op, d_in, n, k, d_out);     \
    } while (0)
    if (bits == 2) { if (nw == 1) TL(1, 2); else if (nw == 2) TL(2, 2); else if (nw == 3) TL(3, 2); else TL(4, 2); }
    else           { if (nw == 1) TL(1, 4); else if (nw == 2) TL(2, 4); else if (nw == 3) TL(3, 4); else TL(4, 4); }
#undef TL
    HIP_TRY(ctx, hipGetLastError());
    if (flags & KMERS_ASYNC) return KMERS_OK;
    if (!dev) HIP_TRY(ctx, hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return KMERS_OK;
}

int kmers_synth_dna(kmers_ctx *ctx, uint64_t seed, uint64_t first_word, uint64_t n_words, int bits,
                    uint32_t ambig_per_65536, uint64_t *out_dev) {
    if (!ctx) return KMERS_E_BADARG;
    if ((bits != 2 && bits != 4) || (n_words && !out_dev)) return fail(ctx, KMERS_E_BADARG, "bad synth arguments");
    if (n_words == 0) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dim3 block(256), grid((unsigned)std::min<uint64_t>((n_words + 255) / 256, 256 * 32));
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, ctx->stream, seed, first_word, n_words, bits, ambig_per_65536, out_dev);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
