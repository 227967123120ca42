// memory_api.hip -- device memory behind the C ABI (include/kmers_hip.h, "device memory"): kmers_dev_alloc / kmers_dev_free /
// kmers_memcpy_* for hosts without a HIP binding of their own, and the context's ARENA.
//
// Why an arena: the reference's `collect(CanonicalDNAMers{31}(seq))` allocates one Vector per call; here the outputs of one
// launch are tens of gigabytes, and WHERE the driver places them is worth 3-5 % of the kernel's write rate on MI355X: the same
// launch into two freshly hipMalloc'ed 8 GB arrays runs at 0.79 of 8 TB/s on a fresh box and at 0.82-0.84 into ranges of one
// large block (profiles/r02_tuning.md section 7; counters in profiles/r03_alloc.md).  kmers_arena_reserve makes that block a
// property of the context instead of an accident of the process's allocation history.
#include <hip/hip_runtime.h>

#include "context.hpp"

using namespace kmers;

namespace {

constexpr size_t GRANULE = KMERS_ARENA_GRANULE;

size_t round_up(size_t x) { return (x + GRANULE - 1) / GRANULE * GRANULE; }

// smallest free range that fits (best fit keeps the large ranges whole for the large outputs)
bool arena_take(kmers_arena &a, size_t need, size_t *off_out) {
    auto best = a.free_ranges.end();
    for (auto it = a.free_ranges.begin(); it != a.free_ranges.end(); ++it)
        if (it->second >= need && (best == a.free_ranges.end() || it->second < best->second)) best = it;
    if (best == a.free_ranges.end()) return false;
    const size_t off = best->first, len = best->second;
    a.free_ranges.erase(best);
    if (len > need) a.free_ranges[off + need] = len - need;
    a.used[off] = need;
    *off_out = off;
    return true;
}

void arena_give(kmers_arena &a, size_t off, size_t len) {
    auto next = a.free_ranges.lower_bound(off);
    if (next != a.free_ranges.end() && off + len == next->first) {  // merge with the range behind
        len += next->second;
        next = a.free_ranges.erase(next);
    }
    if (next != a.free_ranges.begin()) {
        auto prev = std::prev(next);
        if (prev->first + prev->second == off) {  // merge with the range in front
            prev->second += len;
            return;
        }
    }
    a.free_ranges[off] = len;
}

}  // namespace

extern "C" {

int kmers_arena_reserve(kmers_ctx *ctx, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    if (ctx->arena.base) return fail(ctx, KMERS_E_BADARG, "kmers_arena_reserve: this context already holds an arena (release it first)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (bytes == 0) {  // default: three quarters of what is free now
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        bytes = free_b / 4 * 3;
    }
    bytes = round_up(bytes);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "kmers_arena_reserve: hipMalloc", e);
    ctx->arena.base = static_cast<char *>(p);
    ctx->arena.bytes = bytes;
    ctx->arena.free_ranges.clear();
    ctx->arena.used.clear();
    ctx->arena.free_ranges[0] = bytes;
    return KMERS_OK;
}

int kmers_arena_release(kmers_ctx *ctx) {
    if (!ctx) return KMERS_E_BADARG;
    if (!ctx->arena.base) return KMERS_OK;
    if (!ctx->arena.used.empty()) return fail(ctx, KMERS_E_BADARG, "kmers_arena_release: blocks of the arena are still allocated");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipFree(ctx->arena.base));
    ctx->arena = kmers_arena();
    return KMERS_OK;
}

int kmers_arena_info(kmers_ctx *ctx, size_t *reserved, size_t *in_use, size_t *largest_free) {
    if (!ctx) return KMERS_E_BADARG;
    size_t used = 0, largest = 0;
    for (const auto &u : ctx->arena.used) used += u.second;
    for (const auto &f : ctx->arena.free_ranges) largest = f.second > largest ? f.second : largest;
    if (reserved) *reserved = ctx->arena.bytes;
    if (in_use) *in_use = used;
    if (largest_free) *largest_free = largest;
    return KMERS_OK;
}

int kmers_dev_alloc(kmers_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KMERS_E_BADARG;
    *out = nullptr;
    if (ctx->arena.base) {
        size_t off = 0;
        if (arena_take(ctx->arena, round_up(bytes ? bytes : 8), &off)) {
            *out = ctx->arena.base + off;
            return KMERS_OK;
        }
    }  // no arena, or no range of it fits: a plain allocation
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc", e);
    return KMERS_OK;
}

int kmers_dev_free(kmers_ctx *ctx, void *p) {
    if (!ctx) return KMERS_E_BADARG;
    if (!p) return KMERS_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // work of this context that still uses the block
    kmers_arena &a = ctx->arena;
    const char *c = static_cast<const char *>(p);
    if (a.base && c >= a.base && c < a.base + a.bytes) {
        auto it = a.used.find((size_t)(c - a.base));
        if (it == a.used.end()) return fail(ctx, KMERS_E_BADARG, "kmers_dev_free: not the start of a block of the arena");
        const size_t off = it->first, len = it->second;
        a.used.erase(it);
        arena_give(a, off, len);
        return KMERS_OK;
    }
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

}  // extern "C"
