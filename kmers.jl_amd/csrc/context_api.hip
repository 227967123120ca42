// context_api.hip -- the context behind the C ABI (include/kmers_hip.h): life cycle, launch tunables, kmers_sync, and the pure
// host arithmetic of the boundary: Kmer geometry (src/kmer.jl:117-137), iterator lengths, the shard plan.
#include "../../include/kmers_hip.h"

#include "api_common.hpp"

using namespace kmers;

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
    ctx->slot = &device_slot(device);
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
    if (hip_stream) {
        ctx->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return KMERS_E_HIP; }
        ctx->own_stream = true;
    }
    if (hipMalloc(&ctx->d_scratch, 64 * 8) != hipSuccess || hipHostMalloc(&ctx->h_result, 128, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&ctx->h_bounce, BOUNCE_IN + BOUNCE_OUT, hipHostMallocDefault) != hipSuccess ||
        (ctx->d_err = reinterpret_cast<unsigned long long *>(ctx->d_scratch + 1)) == nullptr ||
        hipMemsetAsync(ctx->d_scratch, 0, 64 * 8, ctx->stream) != hipSuccess || hipMemsetAsync(ctx->d_err, 0xFF, 8, ctx->stream) != hipSuccess ||
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
    if (ctx->d_layout) (void)hipFree(ctx->d_layout);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        for (auto &e : ctx->pipe_events)
            if (e) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    kmers::pool_detach(ctx);  // the last context of the device that used the class pool returns its memory to the driver
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *kmers_ctx_stream(kmers_ctx *ctx) { return ctx ? ctx->stream : nullptr; }
const char *kmers_last_error(kmers_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "no context"; }

int kmers_ctx_set_param(kmers_ctx *ctx, int param, int64_t value) {
    if (!ctx) return KMERS_E_BADARG;
    if (param == KMERS_PARAM_TILE_KMERS) ctx->tile_kmers = value;
    else if (param == KMERS_PARAM_MAX_GRID) ctx->max_grid = value;
    else if (param == KMERS_PARAM_SUBTILES) ctx->subtiles = value;
    else if (param == KMERS_PARAM_BLOCK_THREADS) ctx->block_threads = value;
    else if (param == KMERS_PARAM_WIDE_NO_TILES) ctx->wide_no_tiles = value;
    else if (param == KMERS_PARAM_SPLIT_ORDER) ctx->split_order = value;
    else if (param == KMERS_PARAM_STAMPS_PTR) ctx->stamps_ptr = value;
    else if (param == KMERS_PARAM_HOST_CHUNKS) ctx->host_chunks = value;
    else if (param == KMERS_PARAM_POOL) ctx->pool_enable = value;
    else if (param == KMERS_PARAM_POOL_SEARCH_GIB) ctx->pool_search_gib = value;
    else if (param == KMERS_PARAM_POOL_MAX_GIB) ctx->pool_max_gib = value;
    else if (param == KMERS_PARAM_POOL_CACHE) ctx->pool_cache = value;
    else if (param == KMERS_PARAM_SKETCH_HOST_ONLY) ctx->sketch_host_only = value != 0;
    else if (param == KMERS_PARAM_BATCH_PASSES) ctx->batch_passes = value;
    else if (param == KMERS_PARAM_BATCH_DENSE) ctx->batch_dense = value;
    else if (param == KMERS_PARAM_SKETCH_BATCH_LDS) ctx->sketch_batch_lds = value;
    else return fail(ctx, KMERS_E_BADARG, "unknown parameter");
    return KMERS_OK;
}

int kmers_last_launch_shape(kmers_ctx *ctx, int *threads, int *tile_kmers, int *split_order) {
    if (!ctx) return KMERS_E_BADARG;
    if (threads) *threads = ctx->last_threads;
    if (tile_kmers) *tile_kmers = ctx->last_tile;
    if (split_order) *split_order = ctx->last_split;
    return KMERS_OK;
}

int kmers_sync(kmers_ctx *ctx, kmers_result *res) {
    if (!ctx) return KMERS_E_BADARG;
    clear(res);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = collect(ctx, res, 0);
    if (ctx->unamb_pending) {
        // the asynchronous kmers_unambiguous of this context: its element count travelled to pinned memory behind the kernel
        ctx->unamb_pending = false;
        if (rc != KMERS_OK) return rc;  // (an EncodeError of a byte source wins: the outputs are unspecified then)
        const uint64_t *h = ctx->h_result + 8;  // [last tile's descriptor, ticket counter, abort flag]
        if (h[2]) return fail(ctx, KMERS_E_HIP, "UnambiguousKmers: a tile never published its count (look-back gave up)");
        const uint64_t total = h[0] & DESCRIPTOR_COUNT_MASK;
        if (res) res->n_out = total;
        if (total > ctx->unamb_capacity) {
            if (res) res->status = KMERS_E_CAPACITY;
            return fail(ctx, KMERS_E_CAPACITY, "output capacity too small");
        }
    }
    return rc;
}

}  // extern "C"
