// context.hpp -- the context behind the C ABI's opaque kmers_ctx, and the error helpers shared by the
// translation units of libkmers_hip.so (kmers_api.hip: iterators and consumers; comm_api.hip: RCCL).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <atomic>

#include "../../include/kmers_hip.h"
#include "class_pool.hpp"

struct kmers_ctx;
struct kmers_device_pool;  // pool_api.hip
// What ONE DEVICE of the process holds for its contexts: the class pool (pool_api.hip).  `mu` serialises attaching, growing and
// releasing on that device only; slots are never destroyed.  `generation` changes whenever a block of the pool comes or goes: a
// launcher that asked where two arrays lie may keep the answer for as long as it stands (no lock on the launch path).
struct kmers_device_slot {
    std::mutex mu;
    kmers_device_pool *pool = nullptr;
    std::atomic<uint64_t> generation{1};
};

struct kmers_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t *d_scratch = nullptr;        // 64 words of device scratch; word 0: reduction result, 1: the error slot, 2-5: kmers_batch (batch_api.hip); zeroed at creation
    unsigned long long *d_err = nullptr;  // = d_scratch + 1: first offending symbol (0-based), ~0 = none
    uint64_t *h_result = nullptr;         // pinned, 16 words: 0..1 mirror of scratch words 0..1 (one small D2H copy per call), 2..7 per-call
                                          // read-backs of the synchronous entry points, 8..10 the asynchronous kmers_unambiguous
    char *h_bounce = nullptr;             // pinned bounce buffer for short host-pointer calls (FASTA-record sized):
                                          // [0, BOUNCE_IN) source words, [BOUNCE_IN, BOUNCE_IN + BOUNCE_OUT) outputs
    // kmers_batch's layout pass (scan_kernels.hpp): [ticket counter, 3 spare words][descriptors: 2 words per segment], zeroed when
    // allocated; the counter only grows (layout_tickets: its value on the host), a descriptor word counts only with this call's epoch
    unsigned long long *d_layout = nullptr;
    size_t layout_segs = 0;
    uint64_t layout_tickets = 0;
    uint32_t layout_epoch = 0;
    uint64_t last_batch_pieces = 0;       // kmers_last_batch_pieces
    uint64_t *d_recent = nullptr;         // MinHash: table of recently appended candidate hashes (RECENT_SLOTS entries)
    void *stage[8] = {};      // 0 source, 1-2 outputs, 3 metadata / scratch, 4-5 recoded stream / flags, 6 tile index, 7 RCCL scratch
    size_t stage_cap[8] = {};
    std::string last_error;
    int64_t tile_kmers = 0;  // 0 = default
    int64_t max_grid = 0;    // 0 = default
    int64_t subtiles = 0;    // KMERS_PARAM_SUBTILES; 0 = default
    int64_t block_threads = 0;  // KMERS_PARAM_BLOCK_THREADS: 64 / 128 / 256 threads per workgroup of the tile kernel; 0 = per shape
    int64_t split_order = 0;     // KMERS_PARAM_SPLIT_ORDER: the tile kernels visit the two halves of their tile range alternately
    int64_t wide_no_tiles = 0;  // KMERS_PARAM_WIDE_NO_TILES: kmers of more than four words skip wide_tile_kernel.hpp
    int64_t stamps_ptr = 0;  // diagnostic builds only (KMERS_PARAM_STAMPS_PTR)
    int n_cus = 256;                // multiProcessorCount
    bool sketch_host_only = false;  // KMERS_PARAM_SKETCH_HOST_ONLY: force the host-feedback MinHash path (tests)
    int64_t batch_passes = 0;       // KMERS_PARAM_BATCH_PASSES (tests, tuning); 0 = default
    int64_t batch_dense = 0;        // KMERS_PARAM_BATCH_DENSE: -1 = kmers_batch never takes the dense tile path (A/B, tests)
    int64_t sketch_batch_lds = 0;   // KMERS_PARAM_SKETCH_BATCH_LDS (tuning); 0 = default
    bool uses_pool = false;        // this context has taken part in the device's class pool (pool_api.hip): counted in its refs
    int64_t pool_enable = 1;       // KMERS_PARAM_POOL: kmers_dev_alloc of KMERS_POOL_MIN_BYTES or more comes from the class pool
    int64_t pool_search_gib = -1;  // KMERS_PARAM_POOL_SEARCH_GIB: how far past a request the pool may grow in search of the classes it wants (-1: sized to the request)
    int64_t pool_max_gib = 0;      // KMERS_PARAM_POOL_MAX_GIB: cap on what the pool holds (0: what the device has)
    int64_t pool_cache = 1;        // KMERS_PARAM_POOL_CACHE: freed blocks stay mapped for the next request of their shape (0: unmapped at once)
    kmers_device_slot *slot = nullptr;  // the device's slot, looked up once (memory_api.hip)
    // what the launcher last asked the pool about a pair of arrays / the halves of one, and the pool generation the answer is for
    struct placement_answer {
        const void *a = nullptr, *b = nullptr;
        size_t bytes_a = 0, bytes_b = 0;
        uint64_t generation = 0;
        float differ = -1.f;
    } placed_pair, placed_halves;
    int call_flags = KMERS_ASYNC;  // flags of the entry point that is running (the launcher must not block inside a KMERS_ASYNC call)
    hipStream_t copy_stream = nullptr;   // host-pointer calls in chunks (iterators_api.hip): the copies to the host, beside the kernels
    hipEvent_t pipe_events[4] = {};      //   ... kernel done [2], chunk copied [2]
    int64_t host_chunks = 0;             // KMERS_PARAM_HOST_CHUNKS: -1 = host-pointer calls never in chunks (A/B, tests)
    int last_threads = 0, last_tile = 0, last_split = 0;  // shape of the most recent tile-kernel launch (kmers_last_launch_shape)
    bool unamb_pending = false;     // an asynchronous kmers_unambiguous has run since the last kmers_sync: its count is in h_result[8..10]
    uint64_t unamb_capacity = 0;
};

namespace kmers {

constexpr uint64_t DESCRIPTOR_COUNT_MASK = (1ull << 62) - 1ull;  // a tile descriptor of unambiguous_kernel.hpp: status in the top two bits

inline int fail(kmers_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(ctx, call)                                             \
    do {                                                               \
        hipError_t e_ = (call);                                        \
        if (e_ != hipSuccess) return fail(ctx, KMERS_E_HIP, #call, e_); \
    } while (0)

// memory_api.hip: the process's slot for a device (created on first use)
kmers_device_slot &device_slot(int device);
// pool_api.hip: a block of the device's class pool (KMERS_E_UNSUPPORTED: no virtual-memory management here -- plain hipMalloc then);
// free it if `p` is one (*handled); leave the pool (the last context out returns everything to the driver).  For the launchers:
// the fraction of 64 relative positions at which two arrays lie in different region classes / at which the two halves of one
// array do (-1: not inside blocks of the pool).
int pool_alloc(kmers_ctx *ctx, size_t bytes, int role, void **out);
int pool_free(kmers_ctx *ctx, void *p, bool *handled);
void pool_detach(kmers_ctx *ctx);
// pool_api.hip: what the pool holds and nobody uses (cached blocks, free handles) goes back to the driver; the bytes returned.
// memory_api.hip: hipMalloc that tries that before it gives up -- for every buffer the library allocates for itself.
size_t pool_release_idle(kmers_ctx *ctx);
hipError_t dev_malloc(kmers_ctx *ctx, void **out, size_t bytes);
float pool_arrays_differ(kmers_ctx *ctx, const void *a, size_t bytes_a, const void *b, size_t bytes_b);
float pool_halves_differ(kmers_ctx *ctx, const void *a, size_t bytes);

// grow-only device staging buffers owned by the context
inline int ensure_stage(kmers_ctx *ctx, int slot, size_t bytes) {
    if (bytes <= ctx->stage_cap[slot]) return KMERS_OK;
    if (ctx->stage[slot]) (void)hipFree(ctx->stage[slot]);
    ctx->stage[slot] = nullptr;
    ctx->stage_cap[slot] = 0;
    size_t cap = bytes + bytes / 8 + 4096;
    hipError_t e = dev_malloc(ctx, &ctx->stage[slot], cap);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc(staging)", e);
    ctx->stage_cap[slot] = cap;
    return KMERS_OK;
}

}  // namespace kmers
