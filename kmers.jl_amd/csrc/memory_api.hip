// memory_api.hip -- device memory behind the C ABI (include/kmers_hip.h, "device memory"): kmers_dev_alloc / kmers_dev_free /
// kmers_memcpy_* for hosts without a HIP binding of their own.
//
// What the reference does here: `collect(CanonicalDNAMers{31}(seq))` allocates one Vector per call (Base.collect over
// src/iterators/CanonicalKmers.jl:199-225).  On MI355X WHERE the outputs of a launch lie decides whether they are written at 6.2 or
// at 7.2 TB/s (profiles/r03_alloc.md, r05_vmm.md), so arrays of KMERS_POOL_MIN_BYTES or more come from the device's CLASS POOL
// (pool_api.hip); everything smaller, and everything the pool cannot serve, is a plain hipMalloc.  ONE placement mechanism: the
// reservation ("arena") of rounds 3-4 and the launcher's run-time shape calibration that went with it are gone (round 6; they
// never beat the pool by more than process noise, profiles/r05_vmm.md section 6).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "context.hpp"

using namespace kmers;

namespace {

// one slot per device and process (kmers_device_slot, context.hpp): the global mutex guards the map only
std::mutex g_registry_mu;
std::map<int, kmers_device_slot *> g_device_slots;

}  // namespace

kmers_device_slot &kmers::device_slot(int device) {
    std::lock_guard<std::mutex> registry(g_registry_mu);
    kmers_device_slot *&slot = g_device_slots[device];
    if (!slot) slot = new kmers_device_slot();
    return *slot;
}

// hipMalloc for the library's own buffers and for what the pool does not serve: when the driver has nothing left, what the pool
// holds but nobody uses (cached blocks, handles it walked past) goes back to the driver first and the allocation is tried again --
// the library must not report "out of memory" over gigabytes idle in its own lists.
hipError_t kmers::dev_malloc(kmers_ctx *ctx, void **out, size_t bytes) {
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    if (pool_release_idle(ctx) == 0) return e;
    e = hipMalloc(out, bytes);
    if (e != hipSuccess) (void)hipGetLastError();
    return e;
}

extern "C" {

int kmers_dev_alloc(kmers_ctx *ctx, size_t bytes, void **out) { return kmers_dev_alloc_role(ctx, bytes, KMERS_ALLOC_DEFAULT, out); }

int kmers_dev_alloc_role(kmers_ctx *ctx, size_t bytes, int role, void **out) {
    if (!ctx || !out) return KMERS_E_BADARG;
    *out = nullptr;
    if (role != KMERS_ALLOC_DEFAULT && role != KMERS_ALLOC_LONE_OUTPUT) return fail(ctx, KMERS_E_BADARG, "unknown allocation role");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->pool_enable > 0 && bytes >= KMERS_POOL_MIN_BYTES) {
        // arrays of a launch come from the device's class pool (pool_api.hip): a block of physical chunks whose region classes
        // differ from those of the block before it at every relative position (by role: the second half from the first)
        const int rc = pool_alloc(ctx, bytes, role, out);
        if (rc == KMERS_OK) return rc;
        // KMERS_E_UNSUPPORTED (no virtual-memory management here) or KMERS_E_NOMEM (the pool could not get the handles: a cap, a
        // device too full for its four calibration handles): a plain allocation may still fit
        if (rc != KMERS_E_UNSUPPORTED && rc != KMERS_E_NOMEM) return rc;
    }
    hipError_t e = dev_malloc(ctx, out, bytes ? bytes : 8);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: hipMalloc", e);
    return KMERS_OK;
}

int kmers_dev_free(kmers_ctx *ctx, void *p) {
    if (!ctx) return KMERS_E_BADARG;
    if (!p) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a block of the pool goes back without a wait: the pool orders its next use behind the work that is queued now (pool_api.hip)
    bool handled = false;
    if (const int rc = pool_free(ctx, p, &handled)) return rc;
    if (handled) return KMERS_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // work of this context that still uses the block
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

// Pinned host memory and enqueue-only copies: what a host needs to overlap the NEXT chunk of a chunk-buffered iterate() with its
// own loop over the current one (julia/KmersHIP.jl, GPUIterator; kmers.jl_amd/host.py, _ChunkPipe).  The copies complete with the
// context's stream (kmers_sync); from pageable memory an "asynchronous" copy is staged by the runtime and blocks the caller.
int kmers_host_alloc(kmers_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) return KMERS_E_BADARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 8, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, KMERS_E_NOMEM, "kmers_host_alloc: hipHostMalloc", e);
    return KMERS_OK;
}

int kmers_host_free(kmers_ctx *ctx, void *p) {
    if (!ctx) return KMERS_E_BADARG;
    if (!p) return KMERS_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a copy of this context may still be writing it
    HIP_TRY(ctx, hipHostFree(p));
    return KMERS_OK;
}

int kmers_memcpy_h2d_async(kmers_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return KMERS_OK;
}

int kmers_memcpy_d2h_async(kmers_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx) return KMERS_E_BADARG;
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return KMERS_OK;
}

}  // extern "C"
