// context.hpp -- the context behind the C ABI's opaque kmers_ctx, and the error helpers shared by the
// translation units of libkmers_hip.so (kmers_api.hip: iterators and consumers; comm_api.hip: RCCL).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "../../include/kmers_hip.h"

// The context's device-memory arena (kmers_arena_reserve, include/kmers_hip.h): ONE hipMalloc, sub-allocated in 2 MiB
// granules by kmers_dev_alloc.  Offsets are relative to `base`; free ranges are kept coalesced.  `region` is the map of the
// block that the calibration of memory_api.hip measured: two store streams inside one REGION CLASS of HBM share a write rate of
// ~6 TB/s on MI355X, streams in different classes reach ~7.1 TB/s, so consecutive allocations go to different classes.
struct kmers_arena {
    char *base = nullptr;
    size_t bytes = 0;
    std::map<size_t, size_t> free_ranges;  // offset -> length
    std::map<size_t, size_t> used;         // offset -> length
    size_t region_bytes = 0;               // granule of the region map (0: not calibrated)
    std::vector<uint8_t> region;           // class of every granule of the block (kmers_arena_regions)
    std::vector<size_t> run_start;         // the map as runs: run i = [run_start[i], run_start[i + 1]) is in class run_class[i];
    std::vector<uint8_t> run_class;        //   boundaries refined to about half a gigabyte
    std::vector<float> pair_rate;          // measured: pair_rate[i * n_runs + j] = GB/s of two store streams, one in run i, one in run j
    float best_pair_rate = 0.f;            // the largest of them
    int n_classes = 0;
    int last_run = -1, last2_run = -1;     // runs of the two most recent allocations
    size_t last_off = 0, last_len = 0;     // the most recent allocation itself (placement of blocks longer than a run)
};

// index of the arena's run that holds offset `off` (the map must exist)
inline size_t kmers_arena_run_of(const kmers_arena &a, size_t off) {
    size_t lo = 0, hi = a.run_start.size();
    while (hi - lo > 1) {
        const size_t mid = (lo + hi) / 2;
        if (a.run_start[mid] <= off) lo = mid;
        else hi = mid;
    }
    return lo;
}
// true iff two arrays of `bytes` bytes at p and q both lie in the arena and the MEASURED two-stream rate of the runs they pass
// through side by side (sampled at eight points: an array may be longer than a run) averages within 5 % of the best pair of
// the block: the launchers pick the launch shape that is fastest for well-placed outputs only then (stream_launch.hpp)
inline bool kmers_arena_spread(const kmers_arena &a, const void *p, const void *q, size_t bytes) {
    if (a.run_start.empty() || !p || !q || bytes == 0) return false;
    const char *cp = static_cast<const char *>(p), *cq = static_cast<const char *>(q);
    if (cp < a.base || cp + bytes > a.base + a.bytes || cq < a.base || cq + bytes > a.base + a.bytes) return false;
    const size_t k = a.run_start.size();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) {
        const size_t t = (size_t)((2 * i + 1) * (double)bytes / 16.0);
        sum += a.pair_rate[kmers_arena_run_of(a, (size_t)(cp - a.base) + t) * k + kmers_arena_run_of(a, (size_t)(cq - a.base) + t)];
    }
    return sum / 8.f >= 0.95f * a.best_pair_rate;
}

struct kmers_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t *d_scratch = nullptr;        // 64 words of device scratch; word 0: reduction result, word 1: the error slot
    unsigned long long *d_err = nullptr;  // = d_scratch + 1: first offending symbol (0-based), ~0 = none
    uint64_t *h_result = nullptr;         // pinned, 16 words: 0..1 mirror of scratch words 0..1 (one small D2H copy per call), 2..7 per-call
                                          // read-backs of the synchronous entry points, 8..10 the asynchronous kmers_unambiguous
    char *h_bounce = nullptr;             // pinned bounce buffer for short host-pointer calls (FASTA-record sized):
                                          // [0, BOUNCE_IN) source words, [BOUNCE_IN, BOUNCE_IN + BOUNCE_OUT) outputs
    uint64_t *d_recent = nullptr;         // MinHash: table of recently appended candidate hashes (RECENT_SLOTS entries)
    void *stage[8] = {};      // 0 source, 1-2 outputs, 3 metadata / scratch, 4-5 recoded stream / flags, 6 tile index, 7 RCCL scratch
    size_t stage_cap[8] = {};
    std::string last_error;
    int64_t tile_kmers = 0;  // 0 = default
    int64_t max_grid = 0;    // 0 = default
    int64_t subtiles = 0;    // KMERS_PARAM_SUBTILES; 0 = default
    int64_t block_threads = 0;  // KMERS_PARAM_BLOCK_THREADS: 64 / 128 / 256 threads per workgroup of the tile kernel; 0 = per shape
    int64_t split_order = 0;     // KMERS_PARAM_SPLIT_ORDER: the tile kernels visit the two halves of their tile range alternately
    int64_t arena_no_probe = 0;  // KMERS_PARAM_ARENA_NO_PROBE: kmers_arena_reserve skips the region calibration
    int64_t stamps_ptr = 0;  // diagnostic builds only (KMERS_PARAM_STAMPS_PTR)
    int n_cus = 256;                // multiProcessorCount
    bool sketch_host_only = false;  // KMERS_PARAM_SKETCH_HOST_ONLY: force the host-feedback MinHash path (tests)
    int64_t batch_passes = 0;       // KMERS_PARAM_BATCH_PASSES (tests, tuning); 0 = default
    int64_t sketch_batch_lds = 0;   // KMERS_PARAM_SKETCH_BATCH_LDS (tuning); 0 = default
    kmers_arena arena;              // memory_api.hip
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

// grow-only device staging buffers owned by the context
inline int ensure_stage(kmers_ctx *ctx, int slot, size_t bytes) {
    if (bytes <= ctx->stage_cap[slot]) return KMERS_OK;
    if (ctx->stage[slot]) (void)hipFree(ctx->stage[slot]);
    ctx->stage[slot] = nullptr;
    ctx->stage_cap[slot] = 0;
    size_t cap = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc(&ctx->stage[slot], cap);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc(staging)", e);
    ctx->stage_cap[slot] = cap;
    return KMERS_OK;
}

}  // namespace kmers
