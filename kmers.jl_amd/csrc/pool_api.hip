// pool_api.hip -- the device's STRIPED POOL behind kmers_dev_alloc (include/kmers_hip.h, "device memory"): physical memory in
// 32 MiB handles of HIP's virtual-memory management, every handle's HBM region class MEASURED once, every block assembled from
// handles of alternating classes.  The pure logic (which chunks, in which order) is csrc/stripe_pool.hpp; the measurements behind
// the design are profiles/r05_vmm.md (tools/vmm_va.hip, vmm_stripes.hip, vmm_life.hip).
//
// What the reference does here: `collect(CanonicalDNAMers{31}(seq))` allocates one Vector per call (src/iterators/CanonicalKmers.jl:
// 199-225 yields the elements; Base.collect makes the array).  On this device WHERE such an array lies is worth 15 % of the rate
// it can be written at, and rounds 3-4 bought that with a reservation of most of HBM (the arena, memory_api.hip).  The pool needs
// no reservation: it grows by 1 GiB units as blocks are asked for and holds what it was asked for (plus what it had to walk
// past to find a second class, until kmers_pool_trim).
//
// Three properties of the VMM calls on this stack (ROCm 7.2, measured by tools/vmm_life.hip) shape the code:
//   * hipMemMap takes whole handles (no offset): a stripe IS a handle.
//   * A range that is unmapped and mapped again -- in place, or after hipMemAddressFree and a new reservation that returns the
//     same address -- keeps its OLD translations in the device's TLB until something flushes it; a hipMalloc + hipFree does
//     (the legacy unmap goes through KFD, which invalidates).  Every unmap here is followed by that flush, and every new block is
//     checked: a tag written through the chunk's home mapping must be readable through the block.
//   * Physical memory returns to the driver only when the RESERVATION it was mapped under is freed (unmap + release alone keep it).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "context.hpp"

using namespace kmers;
using namespace kmers::pool;

struct kmers_device_pool {
    State s;
    int device = 0;
    int refs = 0;
    hipStream_t stream = nullptr;  // the probes' own stream
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipMemAllocationProp prop = {};
    hipMemAccessDesc access = {};
    bool warm = false;
    uint64_t tag = 0x6b6d657273000000ull;
    size_t n_probes = 0;
};

namespace {

constexpr size_t HALF = UNIT_BYTES / 2;
constexpr float SAME_CLASS = 0.89f;  // a probe at this fraction of the slowest probe's time or more ran inside ONE class (one class:
                                     // 0.94-1.0 of the slowest, two classes: 0.84-0.85; profiles/r05_vmm.md)

// two store streams, 8 KiB of each per workgroup, 16 bytes per lane: the shape of the stream kernels' outputs
__global__ __launch_bounds__(256) void pool_probe_kernel(ulonglong2 *a, ulonglong2 *b) {
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 512u, *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        p[i] = make_ulonglong2(w, i);
        q[i] = make_ulonglong2(i, w);
    }
}

// milliseconds of two streams of HALF bytes at a and b side by side, the best of three
bool probe_ms(kmers_device_pool *P, char *a, char *b, float *out) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        if (hipEventRecord(P->e0, P->stream) != hipSuccess) return false;
        hipLaunchKernelGGL(pool_probe_kernel, dim3((unsigned)(HALF / 8192)), dim3(256), 0, P->stream, reinterpret_cast<ulonglong2 *>(a),
                           reinterpret_cast<ulonglong2 *>(b));
        float ms = 0.f;
        if (hipEventRecord(P->e1, P->stream) != hipSuccess || hipEventSynchronize(P->e1) != hipSuccess ||
            hipEventElapsedTime(&ms, P->e0, P->e1) != hipSuccess)
            return false;
        best = std::min(best, ms);
    }
    ++P->n_probes;
    *out = best;
    return true;
}

// the device's TLB may hold translations of ranges that were just unmapped (header comment): the legacy allocation path flushes it
void flush_tlb() {
    void *p = nullptr;
    if (hipMalloc(&p, CHUNK_BYTES) == hipSuccess) (void)hipFree(p);
    else (void)hipGetLastError();
}

// one more unit: UNIT_CHUNKS handles, mapped side by side under a reservation of their own (the unit's home), each half
// classified against the representatives.  false: the device has no more memory to give (nothing is left half-made).
bool grow_unit(kmers_device_pool *P) {
    State &s = P->s;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    hs.reserve(UNIT_CHUNKS);
    auto undo = [&](char *home, uint32_t mapped) {
        for (uint32_t i = 0; i < mapped; ++i) (void)hipMemUnmap(home + (size_t)i * CHUNK_BYTES, CHUNK_BYTES);
        for (auto h : hs) (void)hipMemRelease(h);
        if (home) (void)hipMemAddressFree(home, UNIT_BYTES);
        if (mapped) flush_tlb();
        (void)hipGetLastError();
    };
    for (uint32_t i = 0; i < UNIT_CHUNKS; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, CHUNK_BYTES, &P->prop, 0) != hipSuccess) {
            undo(nullptr, 0);
            return false;
        }
        hs.push_back(h);
    }
    void *home_v = nullptr;
    if (hipMemAddressReserve(&home_v, UNIT_BYTES, CHUNK_BYTES, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMemAddressReserve(&home_v, UNIT_BYTES, 0, nullptr, 0) != hipSuccess) {
            undo(nullptr, 0);
            return false;
        }
    }
    char *home = static_cast<char *>(home_v);
    for (uint32_t i = 0; i < UNIT_CHUNKS; ++i)
        if (hipMemMap(home + (size_t)i * CHUNK_BYTES, CHUNK_BYTES, 0, hs[i], 0) != hipSuccess) {
            undo(home, i);
            return false;
        }
    if (hipMemSetAccess(home, UNIT_BYTES, &P->access, 1) != hipSuccess) {
        undo(home, UNIT_CHUNKS);
        return false;
    }
    const uint32_t u = (uint32_t)s.units.size(), first = (uint32_t)s.chunks.size();
    Unit unit;
    unit.home = home;
    unit.first_chunk = first;
    s.units.push_back(unit);
    for (uint32_t i = 0; i < UNIT_CHUNKS; ++i) {
        Chunk c;
        c.handle = hs[i];
        c.unit = u;
        s.chunks.push_back(c);
    }
    s.held_bytes += UNIT_BYTES;
    if (!P->warm) {  // the first launches of a process find the clocks idle
        float t;
        for (int i = 0; i < 3; ++i) (void)probe_ms(P, home, home + HALF, &t);
        P->warm = true;
    }
    // classify the two halves (a failed probe leaves them unknown: still usable memory)
    const bool debug = std::getenv("KMERS_POOL_DEBUG") != nullptr;
    char dbg[256];
    int dbg_n = 0;
    dbg[0] = 0;
    for (uint32_t h = 0; h < 2; ++h) {
        char *ptr = home + (size_t)h * HALF;
        int label = -1;
        bool missing = false;
        if (s.n_classes == 0) {  // the very first half IS class A; the probe of the unit's two halves sets the scale
            float t = 0.f;
            if (probe_ms(P, home, home + HALF, &t)) {
                s.slow_ms = s.fast_ms = t;
                label = 0;
                s.rep_ptr[0] = ptr;
                s.rep_unit[0] = u;
                s.rep_half[0] = h;
                s.n_classes = 1;
            }
        } else {
            for (int c = 0; c < s.n_classes && label < 0; ++c) {
                float t = 0.f;
                if (!probe_ms(P, ptr, s.rep_ptr[c], &t)) {
                    missing = true;
                    break;
                }
                if (t > 1.10f * s.slow_ms) {  // slower than anything so far: a hiccup, or the scale was set by a mixed unit -- ask again
                    float t2 = 0.f;
                    if (probe_ms(P, ptr, s.rep_ptr[c], &t2)) t = std::min(t, t2);
                }
                if (debug && dbg_n < 200) dbg_n += std::snprintf(dbg + dbg_n, sizeof dbg - (size_t)dbg_n, " %c%c:%.1f", h ? 'h' : 'l', 'A' + c, 1e3 * t);
                s.slow_ms = std::max(s.slow_ms, t);
                s.fast_ms = std::min(s.fast_ms, t);
                if (t >= SAME_CLASS * s.slow_ms) label = c;
            }
            if (label < 0 && !missing && s.n_classes < MAX_CLASSES) {  // fast beside every representative: a new class
                label = s.n_classes++;
                s.rep_ptr[label] = ptr;
                s.rep_unit[label] = u;
                s.rep_half[label] = h;
            }
        }
        const uint8_t cls = label < 0 ? CLASS_UNKNOWN : (uint8_t)label;
        const bool is_rep = label >= 0 && s.rep_ptr[label] == ptr;
        for (uint32_t i = 0; i < UNIT_CHUNKS / 2; ++i) {
            const uint32_t id = first + h * (UNIT_CHUNKS / 2) + i;
            s.chunks[id].cls = cls;
            s.chunks[id].rep = is_rep;
        }
        // into the free list, in DEscending order so that the list hands them out in creation order; a representative stays out:
        // it is what later units are compared with, and the probes write into it
        for (uint32_t i = UNIT_CHUNKS / 2; !is_rep && i-- > 0;) s.free_list[cls].push_back(first + h * (UNIT_CHUNKS / 2) + i);
    }
    if (debug)
        std::fprintf(stderr, "pool unit %3u: %c %c  (slow %.1f us, fast %.1f us; probes%s)\n", u, 'A' + s.chunks[first].cls, 'A' + s.chunks[first + UNIT_CHUNKS / 2].cls,
                     1e3 * s.slow_ms, 1e3 * s.fast_ms, dbg);
    return true;
}

// give a unit's memory back to the driver (all of its chunks are free and out of the free lists)
void release_unit(kmers_device_pool *P, uint32_t u) {
    State &s = P->s;
    Unit &unit = s.units[u];
    for (uint32_t i = 0; i < UNIT_CHUNKS; ++i) {
        (void)hipMemUnmap(unit.home + (size_t)i * CHUNK_BYTES, CHUNK_BYTES);
        (void)hipMemRelease(static_cast<hipMemGenericAllocationHandle_t>(s.chunks[unit.first_chunk + i].handle));
        s.chunks[unit.first_chunk + i].handle = nullptr;
    }
    (void)hipMemAddressFree(unit.home, UNIT_BYTES);
    (void)hipGetLastError();
    unit.released = true;
    unit.home = nullptr;
    s.held_bytes -= UNIT_BYTES;
}

// units without a chunk in use go back to the driver; `all`: the representatives too (the pool forgets its classes with them)
size_t trim(kmers_device_pool *P, bool all) {
    State &s = P->s;
    size_t released = 0;
    for (uint32_t u = 0; u < s.units.size(); ++u) {
        Unit &unit = s.units[u];
        if (unit.released || unit.in_use) continue;
        bool holds_rep = false;
        for (uint32_t i = 0; i < UNIT_CHUNKS; ++i) holds_rep |= s.chunks[unit.first_chunk + i].rep;
        if (holds_rep && !all) continue;
        for (auto &list : s.free_list)
            list.erase(std::remove_if(list.begin(), list.end(), [&](uint32_t id) { return s.chunks[id].unit == u; }), list.end());
        release_unit(P, u);
        released += UNIT_BYTES;
    }
    if (all && s.held_bytes == 0) {  // nothing left: a later block starts a new pool (classes are measured again)
        s.chunks.clear();
        s.units.clear();
        s.n_classes = 0;
        s.slow_ms = s.fast_ms = 0.f;
        for (auto &p : s.rep_ptr) p = nullptr;
    }
    if (released) flush_tlb();
    return released;
}

void unmap_block(char *base, size_t n_chunks) {
    for (size_t i = 0; i < n_chunks; ++i) (void)hipMemUnmap(base + i * CHUNK_BYTES, CHUNK_BYTES);
    (void)hipMemAddressFree(base, n_chunks * CHUNK_BYTES);
    (void)hipGetLastError();
    flush_tlb();
}

kmers_device_pool *attach(kmers_ctx *ctx, kmers_device_slot &slot) {  // slot.mu is held
    if (!slot.pool) {
        kmers_device_pool *P = new (std::nothrow) kmers_device_pool();
        if (!P) return nullptr;
        P->device = ctx->device;
        P->prop.type = hipMemAllocationTypePinned;
        P->prop.location.type = hipMemLocationTypeDevice;
        P->prop.location.id = ctx->device;
        P->access.location = P->prop.location;
        P->access.flags = hipMemAccessFlagsProtReadWrite;
        int vmm = 0;
        if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, ctx->device) != hipSuccess || !vmm ||
            hipStreamCreateWithFlags(&P->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&P->e0) != hipSuccess ||
            hipEventCreate(&P->e1) != hipSuccess) {
            (void)hipGetLastError();
            if (P->e0) (void)hipEventDestroy(P->e0);
            if (P->stream) (void)hipStreamDestroy(P->stream);
            delete P;
            return nullptr;
        }
        slot.pool = P;
    }
    if (!ctx->uses_pool) {
        ctx->uses_pool = true;
        ++slot.pool->refs;
    }
    return slot.pool;
}

}  // namespace

int kmers::pool_alloc(kmers_ctx *ctx, size_t bytes, void **out) {
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = attach(ctx, slot);
    if (!P) return KMERS_E_UNSUPPORTED;  // no virtual-memory management on this device: the caller falls back to hipMalloc
    State &s = P->s;
    const size_t n = chunks_for(bytes);
    const size_t search_limit = ctx->pool_search_gib >= 0 ? (size_t)ctx->pool_search_gib << 30 : (size_t)64 << 30;
    const size_t max_held = ctx->pool_max_gib > 0 ? (size_t)ctx->pool_max_gib << 30 : ~(size_t)0;
    size_t searched = 0;
    for (;;) {
        size_t free_counts[N_LISTS], total = 0;
        for (int i = 0; i < N_LISTS; ++i) total += free_counts[i] = s.free_list[i].size();
        const bool enough = total >= n;
        if (enough && (balanced(free_counts, n) || searched >= search_limit)) break;
        size_t free_b = 0, total_b = 0;
        const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= UNIT_BYTES + ((size_t)1 << 30) && s.held_bytes + UNIT_BYTES <= max_held;
        if (!room || !grow_unit(P)) {
            (void)hipGetLastError();
            if (enough) break;
            return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: the device has no memory left for the pool");
        }
        if (enough) searched += UNIT_BYTES;
    }
    std::vector<uint32_t> ids = take(s, n);
    if (ids.size() != n) return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: pool bookkeeping");
    void *va = nullptr;
    if (hipMemAddressReserve(&va, n * CHUNK_BYTES, CHUNK_BYTES, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMemAddressReserve(&va, n * CHUNK_BYTES, 0, nullptr, 0) != hipSuccess) {
            give(s, ids);
            return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: hipMemAddressReserve");
        }
    }
    char *base = static_cast<char *>(va);
    hipError_t e = hipSuccess;
    size_t mapped = 0;
    for (; mapped < n && e == hipSuccess; ++mapped)
        e = hipMemMap(base + mapped * CHUNK_BYTES, CHUNK_BYTES, 0, static_cast<hipMemGenericAllocationHandle_t>(s.chunks[ids[mapped]].handle), 0);
    if (e != hipSuccess) --mapped;
    if (e == hipSuccess) e = hipMemSetAccess(base, n * CHUNK_BYTES, &P->access, 1);
    // the block must show the chunks it was made of: first and last chunk, a tag written through the home mapping
    for (int end = 0; end < 2 && e == hipSuccess; ++end) {
        const size_t k = end ? n - 1 : 0;
        const Chunk &c = s.chunks[ids[k]];
        char *home = s.units[c.unit].home + (size_t)(ids[k] - s.units[c.unit].first_chunk) * CHUNK_BYTES;
        const uint64_t tag = ++P->tag;
        uint64_t seen = 0;
        e = hipMemcpy(home, &tag, 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(&seen, base + k * CHUNK_BYTES, 8, hipMemcpyDeviceToHost);
        if (e == hipSuccess && seen != tag) {  // a stale translation: flush and look again
            flush_tlb();
            e = hipMemcpy(&seen, base + k * CHUNK_BYTES, 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess && seen != tag) {
                unmap_block(base, n);
                give(s, ids);
                return fail(ctx, KMERS_E_HIP, "kmers_dev_alloc: a new block of the pool does not show the memory it was mapped to (stale translations)");
            }
        }
    }
    if (e != hipSuccess) {
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap(base + i * CHUNK_BYTES, CHUNK_BYTES);
        (void)hipMemAddressFree(base, n * CHUNK_BYTES);
        flush_tlb();
        give(s, ids);
        return fail(ctx, KMERS_E_HIP, "kmers_dev_alloc: mapping a block of the pool", e);
    }
    Block b;
    b.bytes = n * CHUNK_BYTES;
    b.alternation = alternation_of(s, ids);
    b.chunks = std::move(ids);
    s.blocks[base] = std::move(b);
    *out = base;
    return KMERS_OK;
}

int kmers::pool_free(kmers_ctx *ctx, void *p, bool *handled) {
    *handled = false;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = slot.pool;
    if (!P) return KMERS_OK;
    auto it = P->s.blocks.find(static_cast<const char *>(p));
    if (it == P->s.blocks.end()) {
        if (block_of(P->s, p, 1)) {
            *handled = true;
            return fail(ctx, KMERS_E_BADARG, "kmers_dev_free: not the start of a block of the pool");
        }
        return KMERS_OK;
    }
    *handled = true;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());  // whatever stream of whatever context still writes into it
    unmap_block(static_cast<char *>(p), it->second.chunks.size());
    give(P->s, it->second.chunks);
    P->s.blocks.erase(it);
    return KMERS_OK;
}

void kmers::pool_detach(kmers_ctx *ctx) {
    if (!ctx->uses_pool) return;
    ctx->uses_pool = false;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = slot.pool;
    if (!P || --P->refs > 0) return;
    // the last context of the device that used the pool: everything goes back, blocks that are still out included
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (auto &b : P->s.blocks) {
        unmap_block(const_cast<char *>(b.first), b.second.chunks.size());
        give(P->s, b.second.chunks);
    }
    P->s.blocks.clear();
    trim(P, true);
    (void)hipEventDestroy(P->e0);
    (void)hipEventDestroy(P->e1);
    (void)hipStreamDestroy(P->stream);
    slot.pool = nullptr;
    delete P;
}

float kmers::pool_alternation(kmers_ctx *ctx, const void *p, size_t bytes) {
    if (!ctx->uses_pool || !p || !bytes) return -1.f;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.pool) return -1.f;
    const Block *b = block_of(slot.pool->s, p, bytes);
    return b ? b->alternation : -1.f;
}

extern "C" {

int kmers_pool_info(kmers_ctx *ctx, size_t *held, size_t *in_use, int *n_classes, size_t *class_bytes, double *two_class_gbps, double *one_class_gbps) {
    if (!ctx) return KMERS_E_BADARG;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    const kmers_device_pool *P = slot.pool;
    if (held) *held = P ? P->s.held_bytes : 0;
    if (in_use) *in_use = P ? P->s.in_use_bytes : 0;
    if (n_classes) *n_classes = P ? P->s.n_classes : 0;
    if (class_bytes)
        for (int c = 0; c < KMERS_POOL_CLASSES; ++c) {
            size_t nb = 0;
            if (P)
                for (const Chunk &ch : P->s.chunks) nb += ch.handle && ch.cls == c ? CHUNK_BYTES : 0;
            class_bytes[c] = nb;
        }
    // GB/s of two store streams side by side, as the probes measured them (512 MiB each)
    if (two_class_gbps) *two_class_gbps = P && P->s.n_classes >= 2 && P->s.fast_ms > 0.f ? 2.0 * (double)HALF / 1e6 / (double)P->s.fast_ms : 0.0;
    if (one_class_gbps) *one_class_gbps = P && P->s.slow_ms > 0.f ? 2.0 * (double)HALF / 1e6 / (double)P->s.slow_ms : 0.0;
    return KMERS_OK;
}

int kmers_pool_trim(kmers_ctx *ctx, size_t *released) {
    if (!ctx) return KMERS_E_BADARG;
    if (released) *released = 0;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.pool) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t r = trim(slot.pool, slot.pool->s.blocks.empty());
    if (released) *released = r;
    return KMERS_OK;
}

int kmers_pool_layout(kmers_ctx *ctx, const void *block, size_t *chunk_bytes, unsigned char *classes, size_t capacity, size_t *n_chunks) {
    if (!ctx) return KMERS_E_BADARG;
    if (chunk_bytes) *chunk_bytes = CHUNK_BYTES;
    if (n_chunks) *n_chunks = 0;
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.pool) return KMERS_OK;
    const char *base = nullptr;
    const Block *b = block ? block_of(slot.pool->s, block, 1, &base) : nullptr;
    if (!b) return KMERS_OK;
    if (n_chunks) *n_chunks = b->chunks.size();
    if (classes)
        for (size_t i = 0; i < b->chunks.size() && i < capacity; ++i) classes[i] = slot.pool->s.chunks[b->chunks[i]].cls;
    return KMERS_OK;
}

// What pool_api.hip's header comment says about the VMM calls, checked on THIS box: a chunk mapped under a fresh reservation, the
// reservation unmapped and freed, the TLB flushed, another chunk mapped under the next reservation (which gets the same address
// more often than not): the second chunk's tag must be what is read there.  *stale_without_flush reports what happens without
// the flush (1: the old chunk is still seen -- the behaviour the flush exists for; 0: not reproduced in this run).
int kmers_pool_selftest(kmers_ctx *ctx, int *stale_without_flush) {
    if (!ctx) return KMERS_E_BADARG;
    if (stale_without_flush) *stale_without_flush = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kmers_device_slot &slot = device_slot(ctx->device);
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = attach(ctx, slot);
    if (!P) return fail(ctx, KMERS_E_UNSUPPORTED, "no virtual-memory management on this device");
    State &s = P->s;
    size_t total = 0;
    for (auto &l : s.free_list) total += l.size();
    if (total < 2 && !grow_unit(P)) return fail(ctx, KMERS_E_NOMEM, "kmers_pool_selftest: no memory");
    std::vector<uint32_t> ids = take(s, 2);
    if (ids.size() != 2) return fail(ctx, KMERS_E_NOMEM, "kmers_pool_selftest: two free chunks");
    auto home_of = [&](uint32_t id) { return s.units[s.chunks[id].unit].home + (size_t)(id - s.units[s.chunks[id].unit].first_chunk) * CHUNK_BYTES; };
    int rc = KMERS_OK;
    for (int with_flush = 0; with_flush < 2 && rc == KMERS_OK; ++with_flush) {
        uint64_t tags[2] = {++P->tag, ++P->tag}, seen[2] = {0, 0};
        void *va[2] = {nullptr, nullptr};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipMemcpy(home_of(ids[k]), &tags[k], 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemAddressReserve(&va[k], CHUNK_BYTES, CHUNK_BYTES, nullptr, 0);
            if (e == hipSuccess) e = hipMemMap(va[k], CHUNK_BYTES, 0, static_cast<hipMemGenericAllocationHandle_t>(s.chunks[ids[k]].handle), 0);
            if (e == hipSuccess) e = hipMemSetAccess(va[k], CHUNK_BYTES, &P->access, 1);
            if (e == hipSuccess) e = hipMemcpy(&seen[k], va[k], 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemUnmap(va[k], CHUNK_BYTES);
            if (e == hipSuccess) e = hipMemAddressFree(va[k], CHUNK_BYTES);
            if (e == hipSuccess && with_flush) flush_tlb();
        }
        if (e != hipSuccess) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest", e);
        else if (seen[0] != tags[0]) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest: a fresh mapping does not show its chunk");
        else if (seen[1] != tags[1]) {
            if (with_flush) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest: stale translation in spite of the flush");
            else if (stale_without_flush) *stale_without_flush = 1;
        }
        flush_tlb();
    }
    give(s, ids);
    return rc;
}

}  // extern "C"
