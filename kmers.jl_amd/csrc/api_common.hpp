// api_common.hpp -- what the translation units of libkmers_hip.so's C ABI share on the host side: argument checking (the
// reference's constructor errors), host <-> HBM staging of a sequence, the read-back of the error slot, and the launcher of
// the tile kernel (stream_kernel.hpp) that several entry-point families use.
//   context_api.hip      context, geometry, shard plan              iterators_api.hip    kmers_fw / kmers_canonical / kmers_spaced
//   unambiguous_api.hip  kmers_unambiguous                          consumers_api.hip    fused consumers (XOR, MinHash, minimizers, composition)
//   batch_api.hip        kmers_batch / _spaced / kmers_minhash_batch elementwise_api.hip  kmers_fx_hash / kmers_transform / kmers_synth_dna
//   memory_api.hip       device memory, the arena                   comm_api.hip         the RCCL exchange of the sharded path
// There is deliberately no CPU compute path anywhere in them.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ascii_tables.hpp"
#include "context.hpp"
#include "stream_kernel.hpp"

namespace kmers {

// Pageable host memory costs HIP about 10-15 us per small copy (internal staging + waits); a call on a
// short sequence with host pointers makes three of them.  Copies that fit go through pinned memory instead.
constexpr size_t BOUNCE_IN = 256 << 10, BOUNCE_OUT = 1 << 20;
constexpr int INTERNAL_OUT_DEVICE = 1 << 16;  // batch_impl: out_a / out_b are device pointers even if the pool is host memory

inline void clear(kmers_result *res) {
    if (res) std::memset(res, 0, sizeof *res);
}

// Common argument checks.  K, J errors mirror the constructors (FwKmers.jl:31-35,
// SpacedKmers.jl:26-32); geometry limits are this library's.
inline int check_common(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, int dst_bits, int flags) {
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
    // (kmers of any width: Kmer{A,K,N} has no bound on N, src/kmer.jl:97-111; what an entry point cannot hold it says itself)
    return KMERS_OK;
}

struct Staged {
    const uint64_t *d_words = nullptr;  // device pointer whose word 0 holds first_bit
    uint64_t first_bit = 0;
};

// (a caller that brings a host sequence up in pieces itself: where it comes from; stage 0 byte b is host byte b of `from`)
struct HostSlice {
    const char *from = nullptr;
    size_t bytes = 0;
};

// Make the sequence words available in HBM.  Host memory: copy the words the view touches -- or, with `defer`, only make room for
// them and say where they are.
inline int stage_sequence(kmers_ctx *ctx, const kmers_seq *seq, int flags, Staged *out, HostSlice *defer = nullptr) {
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
        if (defer) {
            defer->from = static_cast<const char *>(from);
            defer->bytes = nbytes;
        } else {
            if (nbytes && nbytes <= BOUNCE_IN) {
                std::memcpy(ctx->h_bounce, from, nbytes);
                from = ctx->h_bounce;
            }
            if (nbytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], from, nbytes, hipMemcpyHostToDevice, ctx->stream));
        }
        out->d_words = static_cast<const uint64_t *>(ctx->stage[0]);
        out->first_bit = 0;
        return KMERS_OK;
    }
    uint64_t w0 = bit0 >> 6;
    uint64_t w1 = (bit0 + seq->n_bases * (uint64_t)seq->src_bits + 63) >> 6;
    size_t bytes = (size_t)(w1 - w0) * 8;
    if (int rc = ensure_stage(ctx, 0, bytes + 8)) return rc;
    const void *from = seq->words + w0;
    if (defer) {
        defer->from = static_cast<const char *>(from);
        defer->bytes = bytes;
    } else {
        if (bytes && bytes <= BOUNCE_IN) {
            std::memcpy(ctx->h_bounce, from, bytes);
            from = ctx->h_bounce;
        }
        if (bytes) HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], from, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    out->d_words = static_cast<const uint64_t *>(ctx->stage[0]);
    out->first_bit = bit0 & 63u;
    return KMERS_OK;
}

// Wait for the stream and turn the device error slot into a kmers_result.
inline int collect(kmers_ctx *ctx, kmers_result *res, uint64_t n_out, uint64_t *value_out = nullptr) {
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
inline uint32_t ascii_table(kmers_ctx *, int dst_bits, int alphabet) {
    if (alphabet == KMERS_ALPHABET_SYMBOLS) return dst_bits == 4 ? (uint32_t)SYMBOL_TABLE_4BIT : (uint32_t)SYMBOL_TABLE_2BIT;
    return (dst_bits == 4 ? 2u : 0u) + (alphabet != 0 ? 1u : 0u);
}

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- entry points of one translation unit that another one calls ------------------------------------------------------
// iterators_api.hip: the tile kernel in MODE_FW (FwKmers + start indices: UnambiguousKmers over a sequence in which nothing can
// be dropped)
int launch_stream_fw(kmers_ctx *ctx, StreamArgs &a, int src_bits, int dst_bits, int n_words, bool vec_ok);
// unambiguous_api.hip: the XOR mode of the one-pass UnambiguousKmers kernel (kmers_reduce_xor_iter, consumers_api.hip)
int unambiguous_xor(kmers_ctx *ctx, const kmers_seq *seq, int k, int stride, uint64_t *out_value, int flags, kmers_result *res);

}  // namespace kmers
