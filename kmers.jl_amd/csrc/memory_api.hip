// memory_api.hip -- device memory behind the C ABI (include/kmers_hip.h, "device memory"): kmers_dev_alloc / kmers_dev_free /
// kmers_memcpy_* for hosts without a HIP binding of their own, and the context's ARENA.
//
// Why an arena: the reference's `collect(CanonicalDNAMers{31}(seq))` allocates one Vector per call; here the outputs of one
// launch are tens of gigabytes, and WHERE the driver places them is worth 3-5 % of the kernel's write rate on MI355X: the same
// launch into two freshly hipMalloc'ed 8 GB arrays runs at 0.79 of 8 TB/s on a fresh box and at 0.82-0.84 into ranges of one
// large block (profiles/r02_tuning.md section 7; counters in profiles/r03_alloc.md).  kmers_arena_reserve makes that block a
// property of the context instead of an accident of the process's allocation history.
//
// What the placement effect IS (round 3, tools/xcd_affinity.hip, profiles/r03_alloc.md): HBM on this device behaves as a few
// REGION CLASSES of tens of gigabytes each.  Store streams that run side by side inside one class share ~6.0 TB/s (75 % of the
// 8 TB/s peak -- the "achievable" figure of every single-buffer bandwidth test); streams in different classes reach ~7.1 TB/s
// (89 %).  The two output arrays of one launch are two such streams.  kmers_arena_reserve therefore MEASURES the map of its
// block (calibrate(): a two-stream fill between every 4 GiB granule and one representative per class found so far; about
// 0.3 s for 200 GB) and kmers_dev_alloc places a block where the measured two-stream rate beside the live blocks is highest.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "context.hpp"

using namespace kmers;

namespace {

using namespace kmers::arena;  // GRANULE, REGION, round_up, run_of, arena_take / arena_give: arena_placement.hpp (pure host logic)
static_assert(GRANULE == KMERS_ARENA_GRANULE, "arena_placement.hpp and include/kmers_hip.h disagree about the granule");
constexpr size_t PROBE = (size_t)1 << 30;     // bytes per stream of one probe

// one slot per device and process (kmers_device_slot, context.hpp): the global mutex guards the map only
std::mutex g_registry_mu;
std::map<int, kmers_device_slot *> g_device_slots;

// two store streams, 8 KiB of each per workgroup, 16 bytes per lane: the shape of the stream kernels' outputs
__global__ __launch_bounds__(256) void arena_probe_kernel(ulonglong2 *a, ulonglong2 *b) {
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 512u, *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        p[i] = make_ulonglong2(w, i);
        q[i] = make_ulonglong2(i, w);
    }
}

// milliseconds of one probe (the fastest of three): PROBE bytes at offset x and PROBE bytes at offset y of the block
int probe_ms(kmers_ctx *ctx, hipEvent_t e0, hipEvent_t e1, size_t x, size_t y, float *out) {
    char *base = ctx->shared_arena->a.base;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(arena_probe_kernel, dim3((unsigned)(PROBE / 8192)), dim3(256), 0, ctx->stream, reinterpret_cast<ulonglong2 *>(base + x),
                           reinterpret_cast<ulonglong2 *>(base + y));
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    *out = best;
    return KMERS_OK;
}

// The region map of a freshly reserved block (nothing allocated yet: the probes write into it).  Granule g is in the class of
// the first representative r with which it is SLOW (two streams in one class); fast with every representative = a new class.
// "Slow" is calibrated on the block itself: the two halves of one granule are in one class (the median over all granules
// discards the few that straddle a boundary).
int calibrate(kmers_ctx *ctx) {
    kmers_arena &a = ctx->shared_arena->a;
    a.region.clear();
    a.run_start.clear();
    a.run_class.clear();
    a.region_bytes = 0;
    a.n_classes = 0;
    a.pair_rate.clear();
    a.best_pair_rate = 0.f;
    a.one_class_rate = 0.f;
    a.last_run = a.last2_run = -1;
    const size_t n = a.bytes / REGION;
    if (n < 4 || ctx->arena_no_probe) return KMERS_OK;
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    int rc = KMERS_OK;
    std::vector<float> same(n);
    float warm;
    rc = probe_ms(ctx, e0, e1, 0, REGION / 2, &warm);  // (first launch of the kernel, clocks)
    for (size_t g = 0; g < n && rc == KMERS_OK; ++g) rc = probe_ms(ctx, e0, e1, g * REGION, g * REGION + REGION / 2, &same[g]);
    std::vector<uint8_t> cls(n, 0);
    std::vector<size_t> refs;
    if (rc == KMERS_OK) {
        std::vector<float> sorted = same;
        std::sort(sorted.begin(), sorted.end());
        const float slow = sorted[n / 2], threshold = 0.93f * slow;  // different classes run at ~0.85 of the one-class time
        for (size_t g = 0; g < n && rc == KMERS_OK; ++g) {
            int c = -1;
            for (size_t r = 0; r < refs.size() && c < 0 && rc == KMERS_OK; ++r) {
                if (refs[r] == g) {
                    c = (int)r;
                    break;
                }
                float t = 0.f;
                rc = probe_ms(ctx, e0, e1, g * REGION, refs[r] * REGION + REGION / 2, &t);
                if (rc != KMERS_OK) break;
                if (t >= threshold) c = (int)r;
            }
            if (c < 0) {
                if (refs.size() >= 16) {  // noise, not structure: no map
                    refs.clear();
                    break;
                }
                refs.push_back(g);
                c = (int)refs.size() - 1;
            }
            cls[g] = (uint8_t)c;
        }
    }
    // a lone granule between two granules of one class is usually a noisy probe, not a region: asked again, against that class
    if (rc == KMERS_OK && refs.size() >= 2) {
        std::vector<float> sorted = same;
        std::sort(sorted.begin(), sorted.end());
        const float threshold = 0.93f * sorted[n / 2];
        for (size_t g = 1; g + 1 < n && rc == KMERS_OK; ++g) {
            if (cls[g - 1] != cls[g + 1] || cls[g] == cls[g - 1]) continue;
            const size_t r = refs[cls[g - 1]];
            float t0 = 0.f, t1 = 0.f;
            rc = probe_ms(ctx, e0, e1, g * REGION, r * REGION + REGION / 2, &t0);
            if (rc == KMERS_OK) rc = probe_ms(ctx, e0, e1, g * REGION + REGION / 2, r * REGION + REGION / 2, &t1);
            if (rc == KMERS_OK && (t0 >= threshold || t1 >= threshold)) cls[g] = cls[g - 1];
        }
    }
    std::vector<size_t> run_start;
    std::vector<uint8_t> run_class;
    if (rc == KMERS_OK && refs.size() >= 2) {
        // runs of granules, then every boundary bisected to the probe's own resolution: x is in class c iff a stream at x is
        // slow beside a stream at c's representative
        std::vector<float> sorted = same;
        std::sort(sorted.begin(), sorted.end());
        const float threshold = 0.93f * sorted[n / 2];
        for (size_t g = 0; g < n && rc == KMERS_OK; ++g) {
            if (g && cls[g] == cls[g - 1]) continue;
            size_t start = g * REGION;
            if (g) {  // the class of granule g - 1 ends somewhere in (its start, the start of g]
                const size_t r = refs[cls[g - 1]];
                size_t lo = (g - 1) * REGION, hi = g * REGION;
                while (hi - lo > PROBE && rc == KMERS_OK) {
                    const size_t mid = lo + (hi - lo) / 2 / GRANULE * GRANULE;
                    float t = 0.f;
                    rc = probe_ms(ctx, e0, e1, mid, r * REGION + (mid >= r * REGION + REGION / 2 && mid < (r + 1) * REGION ? 0 : REGION / 2), &t);
                    if (rc != KMERS_OK) break;
                    if (t >= threshold) lo = mid;
                    else hi = mid;
                }
                start = hi;
            }
            run_start.push_back(start);
            run_class.push_back(cls[g]);
        }
    }
    // what the placement goes by: the measured rate of every PAIR of runs (a stream at the head of each), not the labels --
    // the classes are not all alike (a pair of them may share more than another pair), and a mislabelled granule is harmless
    std::vector<float> pair;
    const size_t k = run_start.size();
    if (rc == KMERS_OK && k >= 2) {
        pair.assign(k * k, 0.f);
        auto run_len = [&](size_t i) { return (i + 1 < k ? run_start[i + 1] : a.bytes) - run_start[i]; };
        for (size_t i = 0; i < k && rc == KMERS_OK; ++i) {
            for (size_t j = i; j < k && rc == KMERS_OK; ++j) {
                if (run_len(i) < 2 * PROBE || run_len(j) < 2 * PROBE) continue;  // (a sliver: rate 0, never preferred)
                float t;
                rc = probe_ms(ctx, e0, e1, run_start[i], run_start[j] + (i == j ? PROBE : 0), &t);
                pair[i * k + j] = pair[j * k + i] = (float)(2.0 * (double)PROBE / 1e6 / (double)t);  // GB/s
            }
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != KMERS_OK) return rc;
    {
        std::vector<float> sorted = same;
        std::sort(sorted.begin(), sorted.end());
        a.one_class_rate = (float)(2.0 * (double)PROBE / 1e6 / (double)sorted[n / 2]);  // GB/s of the two streams of the median probe
    }
    if (refs.size() >= 2) {
        a.region = cls;
        a.region_bytes = REGION;
        a.n_classes = (int)refs.size();
        a.run_start = run_start;
        a.run_class = run_class;
        a.pair_rate = pair;
        a.best_pair_rate = pair.empty() ? 0.f : *std::max_element(pair.begin(), pair.end());
        if (std::getenv("KMERS_ARENA_DEBUG")) {  // the measured map on stderr
            for (size_t i = 0; i < k; ++i) {
                std::fprintf(stderr, "arena run %2zu: class %c at %7.2f GiB:", i, 'A' + run_class[i], (double)run_start[i] / (double)((size_t)1 << 30));
                for (size_t j = 0; j < k; ++j) std::fprintf(stderr, " %4.0f", (double)pair[i * k + j]);
                std::fprintf(stderr, "\n");
            }
        }
    }
    return KMERS_OK;
}

}  // namespace

kmers_device_slot &kmers::device_slot(int device) {
    std::lock_guard<std::mutex> registry(g_registry_mu);
    kmers_device_slot *&slot = g_device_slots[device];
    if (!slot) slot = new kmers_device_slot();
    return *slot;
}

int kmers::arena_detach(kmers_ctx *ctx, bool force) {
    kmers_device_arena *d = ctx->shared_arena;
    if (!d) return KMERS_OK;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> on_device(slot.mu);
    bool last;
    {
        std::lock_guard<std::mutex> lock(d->mu);
        // the blocks THIS context allocated: a context that leaves takes them with it (its stream is the one that may still
        // write them, and nobody could free them through it afterwards)
        size_t mine = 0;
        for (const auto &o : d->owner) mine += o.second == ctx;
        if (mine && !force) return fail(ctx, KMERS_E_BADARG, "kmers_arena_release: blocks this context took from the arena are still allocated");
        if (mine) {
            (void)hipStreamSynchronize(ctx->stream);
            for (auto it = d->owner.begin(); it != d->owner.end();) {
                if (it->second != ctx) {
                    ++it;
                    continue;
                }
                auto u = d->a.used.find(it->first);
                if (u != d->a.used.end()) {
                    const size_t off = u->first, len = u->second;
                    d->a.used.erase(u);
                    arena_give(d->a, off, len);
                }
                it = d->owner.erase(it);
            }
        }
        last = d->refs == 1;
        --d->refs;
    }
    ctx->shared_arena = nullptr;
    if (last) {
        slot.arena = nullptr;
        (void)hipFree(d->a.base);
        delete d;
    }
    return KMERS_OK;
}

extern "C" {

int kmers_arena_reserve(kmers_ctx *ctx, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    if (ctx->shared_arena) return fail(ctx, KMERS_E_BADARG, "kmers_arena_reserve: this context already holds an arena (release it first)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> on_device(slot.mu);  // (this device only: the probes below take 0.3 s)
    if (slot.arena) {
        // the device has its arena already (another context of this process reserved it): ATTACH -- `bytes` is not a second
        // reservation (kmers_arena_info says what there is)
        std::lock_guard<std::mutex> lock(slot.arena->mu);
        ++slot.arena->refs;
        ctx->shared_arena = slot.arena;
        return KMERS_OK;
    }
    if (bytes == 0) {  // default: three quarters of what is free now
        size_t free_b = 0, total_b = 0;
        HIP_TRY(ctx, hipMemGetInfo(&free_b, &total_b));
        bytes = free_b / 4 * 3;
    }
    bytes = round_up(bytes);
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "kmers_arena_reserve: hipMalloc", e);
    kmers_device_arena *d = new (std::nothrow) kmers_device_arena();
    if (!d) {
        (void)hipFree(p);
        return fail(ctx, KMERS_E_NOMEM, "kmers_arena_reserve: out of host memory");
    }
    d->device = ctx->device;
    d->refs = 1;
    d->a.base = static_cast<char *>(p);
    d->a.bytes = bytes;
    d->a.free_ranges[0] = bytes;
    ctx->shared_arena = d;
    if (int rc = calibrate(ctx)) {  // (a failed probe is a HIP failure: give the block back)
        (void)hipFree(p);
        ctx->shared_arena = nullptr;
        delete d;
        return rc;
    }
    slot.arena = d;
    return KMERS_OK;
}

int kmers_arena_regions(kmers_ctx *ctx, void **base, size_t *region_bytes, unsigned char *classes, size_t capacity, size_t *n_regions) {
    if (!ctx) return KMERS_E_BADARG;
    const kmers_arena &a = ctx->arena();
    if (base) *base = a.base;
    if (region_bytes) *region_bytes = a.region_bytes;
    if (n_regions) *n_regions = a.region.size();
    if (classes)
        for (size_t i = 0; i < a.region.size() && i < capacity; ++i) classes[i] = a.region[i];
    return KMERS_OK;
}

int kmers_arena_release(kmers_ctx *ctx) {
    if (!ctx) return KMERS_E_BADARG;
    if (!ctx->shared_arena) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return arena_detach(ctx, false);
}

int kmers_arena_info(kmers_ctx *ctx, size_t *reserved, size_t *in_use, size_t *largest_free) {
    if (!ctx) return KMERS_E_BADARG;
    size_t used = 0, largest = 0, bytes = 0;
    if (ctx->shared_arena) {
        std::lock_guard<std::mutex> lock(ctx->shared_arena->mu);
        const kmers_arena &a = ctx->shared_arena->a;
        for (const auto &u : a.used) used += u.second;
        for (const auto &f : a.free_ranges) largest = f.second > largest ? f.second : largest;
        bytes = a.bytes;
    }
    if (reserved) *reserved = bytes;
    if (in_use) *in_use = used;
    if (largest_free) *largest_free = largest;
    return KMERS_OK;
}

int kmers_arena_rates(kmers_ctx *ctx, double *best_pair_gbps, double *one_class_gbps) {
    if (!ctx) return KMERS_E_BADARG;
    if (best_pair_gbps) *best_pair_gbps = (double)ctx->arena().best_pair_rate;
    if (one_class_gbps) *one_class_gbps = ctx->arena().run_start.empty() ? 0.0 : (double)ctx->arena().one_class_rate;  // (no map: nothing to price against)
    return KMERS_OK;
}

int kmers_placement_probe(kmers_ctx *ctx, void *a_dev, void *b_dev, size_t bytes, double *gbps) {
    if (!ctx) return KMERS_E_BADARG;
    if (!a_dev || !b_dev || !gbps || bytes < 8192 || ((uintptr_t)a_dev & 15u) || ((uintptr_t)b_dev & 15u))
        return fail(ctx, KMERS_E_BADARG, "kmers_placement_probe: two 16-byte aligned device buffers of at least 8 KiB");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t groups = std::min<size_t>(bytes, (size_t)2 << 30) / 8192;  // at most 2 GiB of each are written
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    float best = 1e30f;
    int rc = KMERS_OK;
    for (int rep = 0; rep < 4 && rc == KMERS_OK; ++rep) {  // (the first one warms up)
        hipError_t e = hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(arena_probe_kernel, dim3((unsigned)groups), dim3(256), 0, ctx->stream, static_cast<ulonglong2 *>(a_dev),
                           static_cast<ulonglong2 *>(b_dev));
        if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) rc = fail(ctx, KMERS_E_HIP, "kmers_placement_probe", e);
        else if (rep && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != KMERS_OK) return rc;
    *gbps = 2.0 * (double)groups * 8192.0 / 1e6 / (double)best;
    return KMERS_OK;
}

int kmers_dev_alloc(kmers_ctx *ctx, size_t bytes, void **out) { return kmers_dev_alloc_role(ctx, bytes, KMERS_ALLOC_DEFAULT, out); }

int kmers_dev_alloc_role(kmers_ctx *ctx, size_t bytes, int role, void **out) {
    if (!ctx || !out) return KMERS_E_BADARG;
    *out = nullptr;
    if (role != KMERS_ALLOC_DEFAULT && role != KMERS_ALLOC_LONE_OUTPUT) return fail(ctx, KMERS_E_BADARG, "unknown allocation role");
    if (ctx->shared_arena) {
        size_t off = 0;
        const size_t need = round_up(bytes ? bytes : 8);
        std::lock_guard<std::mutex> lock(ctx->shared_arena->mu);
        kmers_arena &ar = ctx->shared_arena->a;
        // the only output array of its launches: across a class boundary if one has room (else like any other block)
        if ((role == KMERS_ALLOC_LONE_OUTPUT && need >= ((size_t)64 << 20) && arena_take_straddling(ar, need, &off)) ||
            arena_take(ar, need, &off)) {
            ctx->shared_arena->owner[off] = ctx;
            *out = ar.base + off;
            return KMERS_OK;
        }
    } else if (ctx->pool_enable > 0 && bytes >= KMERS_POOL_MIN_BYTES) {
        // no arena: arrays of a launch come from the device's class pool (pool_api.hip) -- a block of physical chunks whose region
        // classes differ from those of the block before it at every relative position (by role: the second half from the first)
        const int rc = pool_alloc(ctx, bytes, role, out);
        if (rc != KMERS_E_UNSUPPORTED) return rc;
    }  // no arena and no pool, a small block, or no range of the arena fits: a plain allocation
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(out, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "hipMalloc", e);
    return KMERS_OK;
}

int kmers_dev_free(kmers_ctx *ctx, void *p) {
    if (!ctx) return KMERS_E_BADARG;
    if (!p) return KMERS_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // work of this context that still uses the block
    bool handled = false;
    if (const int rc = pool_free(ctx, p, &handled)) return rc;
    if (handled) return KMERS_OK;
    const char *c = static_cast<const char *>(p);
    // the device's arena, whether or not THIS context is attached to it (any context of the device may free a block; the stream
    // of the context that allocated it is waited for, it may still be writing)
    kmers_device_arena *d = ctx->shared_arena;
    if (!d) {
        kmers_device_slot &slot = device_slot(ctx->device);
        std::lock_guard<std::mutex> on_device(slot.mu);
        d = slot.arena;
    }
    if (d && c >= d->a.base && c < d->a.base + d->a.bytes) {
        std::lock_guard<std::mutex> lock(d->mu);
        kmers_arena &a = d->a;
        auto it = a.used.find((size_t)(c - a.base));
        if (it == a.used.end()) return fail(ctx, KMERS_E_BADARG, "kmers_dev_free: not the start of a block of the arena");
        auto own = d->owner.find(it->first);
        if (own != d->owner.end()) {
            if (own->second != ctx) HIP_TRY(ctx, hipStreamSynchronize(own->second->stream));
            d->owner.erase(own);
        }
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
