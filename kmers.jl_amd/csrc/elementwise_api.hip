// elementwise_api.hip -- element-wise operations on arrays of kmers and the synthetic input (include/kmers_hip.h): kmers_fx_hash
// (src/kmer.jl:255-261), kmers_transform (src/transformations.jl:1-41, src/kmer.jl:305-384, src/counting.jl:1-8), kmers_synth_dna.
#include "../../include/kmers_hip.h"

#include "api_common.hpp"
#include "elementwise_kernels.hpp"

using namespace kmers;

extern "C" {

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
    if (nw > 4 && dev) {
        // the any-width kernel computes a result word from two input words: overlapping arrays go through a copy
        const char *ib = reinterpret_cast<const char *>(d_in), *ob = reinterpret_cast<const char *>(d_out);
        if (ib < ob + out_bytes && ob < ib + in_bytes) {
            if (int rc = ensure_stage(ctx, 0, in_bytes + 8)) return rc;
            HIP_TRY(ctx, hipMemcpyAsync(ctx->stage[0], kmers, in_bytes, hipMemcpyDeviceToDevice, ctx->stream));
            d_in = (const uint64_t *)ctx->stage[0];
        }
    }
    const bool vec = aligned16(d_in) && aligned16(d_out) && nw != 3;
    const uint64_t items = (vec && nw == 1) ? (n + 1) / 2 : n;
    dim3 block(256), grid((unsigned)std::min<uint64_t>((items + 255) / 256, (uint64_t)1 << 30));  // one pass per workgroup
#define TL(NW_, B_)                                                                                                      \
    do {                                                                                                                 \
        if (vec) hipLaunchKernelGGL((transform_kernel<NW_, B_, true>), grid, block, 0, ctx->stream, op, d_in, n, k, d_out);  \
        else hipLaunchKernelGGL((transform_kernel<NW_, B_, false>), grid, block, 0, ctx->stream, op, d_in, n, k, d_out);     \
    } while (0)
    if (nw > 4) {  // any width: transform_kernel_any (in and out must not alias: a result word reads two input words)
        grid = dim3((unsigned)std::min<uint64_t>((n + 255) / 256, (uint64_t)1 << 30));
        if (bits == 2) hipLaunchKernelGGL(transform_kernel_any<2>, grid, block, 0, ctx->stream, op, d_in, n, k, nw, d_out);
        else hipLaunchKernelGGL(transform_kernel_any<4>, grid, block, 0, ctx->stream, op, d_in, n, k, nw, d_out);
    } else if (bits == 2) { if (nw == 1) TL(1, 2); else if (nw == 2) TL(2, 2); else if (nw == 3) TL(3, 2); else TL(4, 2); }
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
