// pool_api.hip -- the device's CLASS POOL behind kmers_dev_alloc (include/kmers_hip.h, "device memory"): physical memory in 1 GiB
// handles of HIP's virtual-memory management, the HBM region class of every handle MEASURED once, every block assembled from
// handles of the classes it should have.  The pure logic (which classes where) is csrc/class_pool.hpp; the measurements behind
// the design are profiles/r05_vmm.md (tools/device_probes/vmm_va.hip, vmm_stripes.hip, vmm_life.hip).
//
// What the reference does here: `collect(CanonicalDNAMers{31}(seq))` allocates one Vector per call (src/iterators/CanonicalKmers.jl:
// 199-225 yields the elements; Base.collect makes the array).  On this device WHERE such an array lies is worth 15 % of the rate
// it can be written at, and rounds 3-4 bought that with a reservation of most of HBM (the arena, memory_api.hip) whose map had to
// be coarse enough for whole arrays to fit its runs.  The pool needs no reservation and does not care how finely the classes are
// interleaved in physical memory: it grows by one handle at a time as blocks are asked for and holds what it was asked for plus
// what it had to walk past to find the classes it wanted (until kmers_pool_trim).
//
// Four properties of the VMM calls on this stack (ROCm 7.2, measured by tools/device_probes/vmm_life.hip and vmm_churn.hip) shape
// the code:
//   * hipMemMap takes whole handles (no offset).
//   * A 1 GiB handle mapped ON a GiB boundary faults now and then while handles come and go: no mapping begins on one (reserve()).
//   * A range that is unmapped and mapped again -- in place, or after hipMemAddressFree and a new reservation that returns the
//     same address -- keeps its OLD translations in the device's TLB until something flushes it; a hipMalloc + hipFree does
//     (the legacy unmap goes through KFD, which invalidates).  Every unmap here is followed by that flush, and every new block is
//     checked: a tag written through a chunk's home mapping must be readable through the block.
//   * Physical memory returns to the driver only when the RESERVATION it was mapped under is freed (unmap + release alone keep it).
//
// Round 6 -- what a block COSTS (profiles/r06_pool.md; VERDICT r5 "price the class pool on the collect path").  A host loop is
// {allocate the outputs, launch, consume, free} (Base.collect makes a fresh Vector per call), and round 5's block paid a
// reservation, a map per handle, two blocking 8-byte copies and -- at the free -- a hipDeviceSynchronize, an unmap per handle and a
// TLB flush (a hipMalloc + hipFree), beside a 2.3 ms kernel.  Now:
//   * kmers_dev_free does not wait.  It records an event on the stream of every context of the device that uses the pool and puts
//     the block, STILL MAPPED, into the pool's cache; the next request of its shape takes it as it is (class_pool.hpp,
//     find_cached) and its stream waits for those events -- on the device, not on the host; a request from the stream that
//     freed it waits for nothing.  A loop that allocates the same shapes makes no call into the driver at all.
//   * A block of ONE handle (arrays of KMERS_POOL_MIN_BYTES = 128 MiB up to 1 GiB) is the handle's home mapping: nothing is
//     reserved, mapped or unmapped, ever.
//   * A cached block nobody asked for over CACHE_AGE_TICKS allocations / frees is taken apart (its events waited for, unmapped),
//     and so is the oldest one whenever the pool would otherwise have to grow past what the device has.
//   * The hoard is bounded: after every allocation and free the pool returns to the driver what it holds outside blocks beyond
//     max(4 GiB, a quarter of its blocks) -- the handles it walked past in search of a class are held for the search only.
//   * EVERY handle of a newly mapped block is checked against its home mapping (round 5: the first and the last), and a TLB flush
//     that cannot run is an error, not a shrug.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cmath>
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
    float same_ms = 0.f;  // a probe that takes this long or longer ran inside ONE class: the geometric mean of the two levels below
    float one_ms = 0.f, two_ms = 0.f;  // running estimates of the two levels (one class / two classes), updated by every confident probe
    uint64_t tag = 0x6b6d657273000000ull;
    size_t n_probes = 0;
    std::vector<kmers_ctx *> users;       // the contexts that take part (their streams are what may still write a freed block)
    std::vector<hipEvent_t> spare_events;
    struct Pending {
        hipStream_t stream;
        hipEvent_t ev;
    };
    std::map<const char *, std::vector<Pending>> pending;  // cached block -> the work that was queued when it was freed
    std::vector<uint32_t> dead_ids;  // slots of State::chunks whose handle went back to the driver
    bool need_flush = false;  // something was unmapped since the last TLB flush: nothing may be mapped before the next one
    uint64_t cache_hits = 0, cache_misses = 0, evictions = 0, chunks_created = 0, chunks_returned = 0;
};

namespace {

constexpr float SAME_CLASS = 0.93f;  // of the one-class level while the two-class level is unknown (one class: 0.96-1.0 of it, two classes: 0.82-0.87)
constexpr size_t FLUSH_BYTES = (size_t)32 << 20;

// two store streams, 8 KiB of each per workgroup, 16 bytes per lane: the shape of the stream kernels' outputs
__global__ __launch_bounds__(256) void pool_probe_kernel(ulonglong2 *a, ulonglong2 *b) {
    const uint64_t w = blockIdx.x;
    ulonglong2 *p = a + w * 512u, *q = b + w * 512u;
    for (uint32_t i = threadIdx.x; i < 512u; i += 256u) {
        p[i] = make_ulonglong2(w, i);
        q[i] = make_ulonglong2(i, w);
    }
}

// milliseconds of two streams of one chunk each side by side, the best of three
bool probe_ms(kmers_device_pool *P, char *a, char *b, float *out) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        if (hipEventRecord(P->e0, P->stream) != hipSuccess) return false;
        hipLaunchKernelGGL(pool_probe_kernel, dim3((unsigned)(CHUNK_BYTES / 8192)), dim3(256), 0, P->stream, reinterpret_cast<ulonglong2 *>(a),
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

// the device's TLB may hold translations of ranges that were just unmapped (header comment): the legacy allocation path flushes
// it.  An unmap marks the pool; the flush runs ONCE at the end of the pool operation that unmapped (settle(), below) and in any
// case before the next map.  (It must not wait for the next map: between a hipMemAddressFree and the flush's hipMalloc + hipFree
// the runtime's own books are off -- a plain hipMalloc of the HOST that landed on the freed range aborted inside its hipFree,
// "Memobj map does not have ptr", tests/test_gpu_pool.py on ROCm 7.2 -- so the flush's allocation is the first one to follow.)
// false: the flush could not run (no 32 MiB for its hipMalloc even after the pool's idle handles went back) -- nothing may be mapped.
bool release_idle_locked(kmers_device_pool *P, bool cached_too);
bool flush_tlb(kmers_device_pool *P) {
    if (!P->need_flush) return true;
    for (int attempt = 0; attempt < 2; ++attempt) {
        void *p = nullptr;
        if (hipMalloc(&p, FLUSH_BYTES) == hipSuccess) {
            (void)hipFree(p);
            P->need_flush = false;
            return true;
        }
        (void)hipGetLastError();
        if (attempt == 0 && !release_idle_locked(P, true)) break;  // (returns what is idle; itself ends with need_flush set)
    }
    return false;
}

void settle(kmers_device_pool *P) {
    if (P->need_flush) (void)flush_tlb(P);
}
struct Settle {
    kmers_device_pool *P;
    ~Settle() { settle(P); }
};

// NO MAPPING BEGINS ON A GiB BOUNDARY.  A 1 GiB handle mapped on one becomes a single 1 GiB page: the kernel driver frees the page
// tables that covered the range before, and under the churn of this pool (handles created and returned by the hundred while it
// searches for a class) the first access through such a mapping now and then dies with "Memory access fault by GPU ... Reason:
// Unknown" -- one bench.py run in about twenty in round 6, reproduced in plain HIP by tools/device_probes/vmm_churn.hip: with
// every mapping forced onto a GiB boundary 3 runs of 3 die within 30 s, with none on one 20 runs of 20 / 30 600 handles survive,
// left to the runtime's choice (about one address in thirty is a multiple of 1 GiB) 3 of 11 die (profiles/r06_pool.md section 6).  So every
// reservation is 2 MiB longer than what is mapped under it, and the mapping begins 2 MiB in if the reservation begins on a GiB
// boundary; the handles of a block lie 1 GiB apart, so none of them begins on one either.
constexpr size_t VA_PAD = (size_t)2 << 20;
bool reserve(char **at, size_t bytes, uint32_t *lead) {
    void *va = nullptr;
    if (hipMemAddressReserve(&va, bytes + VA_PAD, VA_PAD, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (hipMemAddressReserve(&va, bytes + VA_PAD, 0, nullptr, 0) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    }
    *lead = reinterpret_cast<uintptr_t>(va) % CHUNK_BYTES == 0 ? (uint32_t)VA_PAD : 0u;
    *at = static_cast<char *>(va) + *lead;
    return true;
}
void unreserve(char *at, size_t bytes, uint32_t lead) {
    (void)hipMemAddressFree(at - lead, bytes + VA_PAD);
    (void)hipGetLastError();
}

// one more handle, mapped under a reservation of its own (its home); not yet in any free list.  false: the device has no
// more memory to give (nothing is left half-made).
bool create_chunk(kmers_device_pool *P, uint32_t *id) {
    if (!flush_tlb(P)) return false;  // (the home reservation may get an address that was unmapped a moment ago)
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, CHUNK_BYTES, &P->prop, 0) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    char *home = nullptr;
    uint32_t lead = 0;
    if (!reserve(&home, CHUNK_BYTES, &lead)) {
        (void)hipMemRelease(h);
        return false;
    }
    if (hipMemMap(home, CHUNK_BYTES, 0, h, 0) != hipSuccess || hipMemSetAccess(home, CHUNK_BYTES, &P->access, 1) != hipSuccess) {
        (void)hipMemUnmap(home, CHUNK_BYTES);
        (void)hipMemRelease(h);
        unreserve(home, CHUNK_BYTES, lead);
        P->need_flush = true;
        return false;
    }
    Chunk c;
    c.handle = h;
    c.home = home;
    c.va_lead = lead;
    if (!P->dead_ids.empty()) {  // (the slot of a handle that went back to the driver: a long-lived process walks many)
        *id = P->dead_ids.back();
        P->dead_ids.pop_back();
        P->s.chunks[*id] = c;
    } else {
        *id = (uint32_t)P->s.chunks.size();
        P->s.chunks.push_back(c);
    }
    P->s.held_bytes += CHUNK_BYTES;
    ++P->chunks_created;
    return true;
}

// give a chunk's memory back to the driver (it is free and out of the free lists); the TLB is flushed before the next map
void destroy_chunk(kmers_device_pool *P, uint32_t id) {
    Chunk &c = P->s.chunks[id];
    (void)hipMemUnmap(c.home, CHUNK_BYTES);
    (void)hipMemRelease(static_cast<hipMemGenericAllocationHandle_t>(c.handle));
    unreserve(c.home, CHUNK_BYTES, c.va_lead);
    c = Chunk();
    P->dead_ids.push_back(id);
    P->s.held_bytes -= CHUNK_BYTES;
    P->need_flush = true;
    ++P->chunks_returned;
}

void debug_chunk(const State &s, uint32_t id, const char *probes) {
    if (std::getenv("KMERS_POOL_DEBUG"))
        std::fprintf(stderr, "pool chunk %3u: %c%s  (one class %.1f us, two %.1f us;%s)\n", id, "ABC?"[s.chunks[id].cls], s.chunks[id].rep ? " (representative)" : "",
                     1e3 * s.slow_ms, 1e3 * s.fast_ms, probes);
}

void set_levels(kmers_device_pool *P, float one_ms, float two_ms) {
    P->one_ms = one_ms;
    P->two_ms = two_ms;
    P->same_ms = two_ms > 0.f ? std::sqrt(one_ms * two_ms) : SAME_CLASS * one_ms;
    P->s.slow_ms = one_ms;
    if (two_ms > 0.f) P->s.fast_ms = two_ms;
}

// The first four handles: every pair probed.  With three classes two of the four share one, so the slow end of the six times is
// two streams inside one class; the widest gap between neighbouring times (if it is 6 % or more) separates the two levels, and
// the MEDIAN of each side is its estimate (a single slow outlier as the yardstick made same-class probes look fast: spurious
// fourth and fifth classes on two boxes of round 5).
bool room_for_chunks(const kmers_ctx *ctx, const State &s, size_t n);
bool calibrate(kmers_device_pool *P, const kmers_ctx *ctx) {
    State &s = P->s;
    uint32_t id[4];
    int made = 0;
    if (!room_for_chunks(ctx, s, 4)) return false;  // (the cap of KMERS_PARAM_POOL_MAX_GIB and the device's free memory hold for the calibration too)
    for (; made < 4; ++made)
        if (!create_chunk(P, &id[made])) break;
    if (made < 4) {
        for (int i = 0; i < made; ++i) destroy_chunk(P, id[i]);
        s.chunks.clear();
        P->dead_ids.clear();
        return false;
    }
    float t[4][4] = {}, warm;
    for (int i = 0; i < 3; ++i) (void)probe_ms(P, s.chunks[id[0]].home, s.chunks[id[1]].home, &warm);  // the first launches of a process find the clocks idle
    std::vector<float> all;
    bool ok = true;
    for (int i = 0; i < 4 && ok; ++i)
        for (int j = i + 1; j < 4 && ok; ++j) {
            ok = probe_ms(P, s.chunks[id[i]].home, s.chunks[id[j]].home, &t[i][j]);
            all.push_back(t[i][j]);
        }
    if (!ok) {  // probes do not run: memory without classes (everything unknown), still a pool
        for (int i = 0; i < 4; ++i) s.free_list[CLASS_UNKNOWN].push_back(id[i]);
        s.slow_ms = s.fast_ms = 0.f;
        P->same_ms = 1e30f;
        return true;
    }
    std::sort(all.begin(), all.end());
    size_t cut = 0;  // all[cut ..] is the slow side
    float widest = 0.f;
    for (size_t i = 1; i < all.size(); ++i) {
        const float gap = all[i] / all[i - 1] - 1.f;
        if (gap > widest) {
            widest = gap;
            cut = i;
        }
    }
    auto median = [&](size_t lo, size_t hi) { return all[lo + (hi - lo - 1) / 2]; };
    if (widest >= 0.06f) set_levels(P, median(cut, all.size()), median(0, cut));
    else set_levels(P, median(0, all.size()), 0.f);  // one level only: the four share a class
    for (int i = 0; i < 4; ++i) {
        int label = -1;
        for (int j = 0; j < i && label < 0; ++j)
            if (t[j][i] >= P->same_ms) label = s.chunks[id[j]].cls;
        Chunk &c = s.chunks[id[i]];
        if (label < 0 && s.n_classes < MAX_CLASSES) {
            label = s.n_classes++;
            s.rep_chunk[label] = id[i];
            c.rep = true;
        } else if (label < 0) {
            label = 0;
        }
        c.cls = (uint8_t)label;
        if (!c.rep) s.free_list[label].push_back(id[i]);
        char buf[96];
        int n = 0;
        buf[0] = 0;
        for (int j = 0; j < i; ++j) n += std::snprintf(buf + n, sizeof buf - (size_t)n, " %u:%.1f", id[j], 1e3 * t[j][i]);
        debug_chunk(s, id[i], buf);
    }
    return true;
}

// one more chunk, probed beside the representative of EVERY class found so far: it belongs to the class it is slowest beside, if
// that probe is on the one-class side; a chunk that is fast beside all of them is a new class while fewer than three are known,
// and otherwise (a handle that straddles a boundary of the physical map, a noisy probe: asked again first) the nearest one.
// Confident probes move the two levels (exponential averages), so the yardstick follows the box.
bool grow(kmers_device_pool *P, const kmers_ctx *ctx) {
    State &s = P->s;
    if (s.chunks.empty() || (s.n_classes == 0 && s.slow_ms == 0.f && P->same_ms == 0.f)) return calibrate(P, ctx);
    uint32_t id;
    if (!create_chunk(P, &id)) return false;
    Chunk &c = s.chunks[id];
    bool failed = P->same_ms >= 1e29f;
    char buf[128];
    int n = 0;
    buf[0] = 0;
    float t[MAX_CLASSES] = {};
    int best = -1;
    for (int k = 0; k < s.n_classes && !failed; ++k) {
        if (!probe_ms(P, c.home, s.chunks[s.rep_chunk[k]].home, &t[k])) {
            failed = true;
            break;
        }
        n += std::snprintf(buf + n, sizeof buf - (size_t)n, " %c:%.1f", 'A' + k, 1e3 * t[k]);
        if (best < 0 || t[k] > t[best]) best = k;
    }
    int label = -1;
    if (!failed && best >= 0) {
        if (t[best] < P->same_ms && s.n_classes >= MAX_CLASSES) {  // like none of the three: ask the nearest again
            float t2 = 0.f;
            if (probe_ms(P, c.home, s.chunks[s.rep_chunk[best]].home, &t2)) t[best] = std::max(t[best], t2);
            n += std::snprintf(buf + n, sizeof buf - (size_t)n, " %c again:%.1f", 'A' + best, 1e3 * t2);
        }
        if (t[best] >= P->same_ms) {
            label = best;
            float one = 0.9f * P->one_ms + 0.1f * t[best], two = P->two_ms;
            for (int k = 0; k < s.n_classes; ++k)
                if (k != best && t[k] < P->same_ms) two = two > 0.f ? 0.9f * two + 0.1f * t[k] : t[k];
            set_levels(P, one, two);
        } else if (s.n_classes < MAX_CLASSES) {  // fast beside every representative: a new class, and its yardstick
            label = s.n_classes++;
            s.rep_chunk[label] = id;
            c.rep = true;
            float two = P->two_ms;
            for (int k = 0; k < label; ++k) two = two > 0.f ? 0.9f * two + 0.1f * t[k] : t[k];
            set_levels(P, P->one_ms, two);
        } else {
            label = best;  // the nearest
        }
    }
    c.cls = label < 0 ? CLASS_UNKNOWN : (uint8_t)label;
    if (!c.rep) s.free_list[c.cls].push_back(id);
    debug_chunk(s, id, buf);
    return true;
}

// ---- events: the work that may still touch a freed block --------------------------------------------------------------------
hipEvent_t get_event(kmers_device_pool *P) {
    if (!P->spare_events.empty()) {
        hipEvent_t e = P->spare_events.back();
        P->spare_events.pop_back();
        return e;
    }
    // No system-scope fence: these events order work on the DEVICE (the next user's stream, or the host before it unmaps) and carry no
    // data to the host; with the fence a record costs the stream 3 us, without it 0.3-1 (tools/device_probes/marker_cost.hip).
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return e;
}

// everything queued so far on the streams of the pool's contexts: what a later user of the block has to come after.  A stream an
// event cannot be recorded on is waited for here and now instead.
void record_pending(kmers_device_pool *P, const char *base) {
    std::vector<kmers_device_pool::Pending> &list = P->pending[base];
    for (kmers_ctx *u : P->users) {
        bool seen = false;
        for (const auto &q : list) seen = seen || q.stream == u->stream;
        if (seen) continue;  // (contexts that borrow one stream)
        hipEvent_t e = get_event(P);
        if (e && hipEventRecord(e, u->stream) == hipSuccess) {
            list.push_back({u->stream, e});
            continue;
        }
        (void)hipGetLastError();
        if (e) P->spare_events.push_back(e);
        (void)hipStreamSynchronize(u->stream);
    }
}
void wait_pending_on_host(kmers_device_pool *P, const char *base) {
    auto it = P->pending.find(base);
    if (it == P->pending.end()) return;
    for (auto &q : it->second) {
        (void)hipEventSynchronize(q.ev);
        P->spare_events.push_back(q.ev);
    }
    (void)hipGetLastError();
    P->pending.erase(it);
}
// the new user's stream comes after them (its own earlier work is before it anyway)
void wait_pending_on_stream(kmers_device_pool *P, const char *base, hipStream_t st) {
    auto it = P->pending.find(base);
    if (it == P->pending.end()) return;
    for (auto &q : it->second) {
        if (q.stream != st && hipStreamWaitEvent(st, q.ev, 0) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipEventSynchronize(q.ev);
        }
        P->spare_events.push_back(q.ev);  // (a wait refers to the record that was current when it was enqueued: the event may be recorded again)
    }
    P->pending.erase(it);
}

// a block's own mapping goes; its chunks stay with the pool.  The TLB is flushed before the next map (flush_tlb).
void unmap_block(kmers_device_pool *P, char *base, const Block &b) {
    if (b.home) return;  // (the chunk's own mapping: it lives as long as the chunk)
    for (size_t i = 0; i < b.chunks.size(); ++i) (void)hipMemUnmap(base + i * CHUNK_BYTES, CHUNK_BYTES);
    unreserve(base, b.chunks.size() * CHUNK_BYTES, b.va_lead);
    P->need_flush = true;
}

// a cached block is taken apart: the work that was queued when it was freed is waited for (on the host: its memory may go
// anywhere next), its chunks return to the free lists
void evict(kmers_device_pool *P, std::map<const char *, Block>::iterator it) {
    wait_pending_on_host(P, it->first);
    unmap_block(P, const_cast<char *>(it->first), it->second);
    give(P->s, it->second.chunks, true);
    P->s.cached.erase(it);
    ++P->evictions;
}
bool evict_oldest(kmers_device_pool *P) {
    State &s = P->s;
    if (s.cached.empty()) return false;
    auto oldest = s.cached.begin();
    for (auto it = s.cached.begin(); it != s.cached.end(); ++it)
        if (it->second.freed_tick < oldest->second.freed_tick) oldest = it;
    evict(P, oldest);
    return true;
}
void age_out(kmers_device_pool *P) {
    State &s = P->s;
    for (auto it = s.cached.begin(); it != s.cached.end();) {
        auto next = std::next(it);
        if (s.tick - it->second.freed_tick > CACHE_AGE_TICKS) evict(P, it);
        it = next;
    }
}

// what the pool holds outside blocks beyond what it may (class_pool.hpp, hoard_excess) goes back to the driver: from the fullest
// free list first, so that what stays is a mix of classes
size_t trim_hoard(kmers_device_pool *P) {
    State &s = P->s;
    size_t released = 0;
    for (size_t x = hoard_excess(s); x > 0; --x) {
        int fullest = -1;
        for (int c = 0; c < N_LISTS; ++c)
            if (!s.free_list[c].empty() && (fullest < 0 || s.free_list[c].size() > s.free_list[fullest].size())) fullest = c;
        if (fullest < 0) break;
        const uint32_t id = s.free_list[fullest].back();
        s.free_list[fullest].pop_back();
        destroy_chunk(P, id);
        released += CHUNK_BYTES;
    }
    return released;
}

// free chunks go back to the driver; `all`: the representatives too (the pool forgets its classes with them)
size_t trim(kmers_device_pool *P, bool all) {
    State &s = P->s;
    size_t released = 0;
    for (auto &list : s.free_list) {
        for (uint32_t id : list) {
            destroy_chunk(P, id);
            released += CHUNK_BYTES;
        }
        list.clear();
    }
    if (all && s.in_use_bytes == 0 && s.cached_bytes == 0) {
        for (int k = 0; k < s.n_classes; ++k) {
            destroy_chunk(P, s.rep_chunk[k]);
            released += CHUNK_BYTES;
        }
        s.chunks.clear();  // a later block starts a new pool (classes are measured again)
        P->dead_ids.clear();
        s.n_classes = 0;
        s.slow_ms = s.fast_ms = 0.f;
        P->same_ms = 0.f;
    }
    return released;
}

// everything nobody uses goes back to the driver: cached blocks (if asked) and free chunks.  true: something was returned.
bool release_idle_locked(kmers_device_pool *P, bool cached_too) {
    bool any = false;
    if (cached_too)
        while (evict_oldest(P)) any = true;
    return trim(P, false) > 0 || any;
}

kmers_device_pool *attach(kmers_ctx *ctx, kmers_device_slot &slot, bool create = true) {  // slot.mu is held
    if (!slot.pool) {
        if (!create) return nullptr;
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
        slot.pool->users.push_back(ctx);
    }
    return slot.pool;
}

// the device has room for n more handles (and one to spare) and the pool's cap allows them
bool room_for_chunks(const kmers_ctx *ctx, const State &s, size_t n) {
    const size_t max_held = ctx->pool_max_gib > 0 ? (size_t)ctx->pool_max_gib << 30 : ~(size_t)0;
    size_t free_b = 0, total_b = 0;
    return hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= (n + 1) * CHUNK_BYTES && s.held_bytes + n * CHUNK_BYTES <= max_held;
}

// every chunk of a newly mapped block must show through the block what was written through its home mapping (a stale translation
// of the address range would show other memory: kernel stores through it would land in somebody else's block)
hipError_t verify_block(kmers_device_pool *P, char *base, const std::vector<uint32_t> &ids, bool *stale) {
    const size_t n = ids.size();
    std::vector<uint64_t> tags(n), seen(n, 0);
    hipError_t e = hipSuccess;
    for (size_t k = 0; k < n && e == hipSuccess; ++k) {
        tags[k] = ++P->tag;
        e = hipMemcpyAsync(P->s.chunks[ids[k]].home, &tags[k], 8, hipMemcpyHostToDevice, P->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(P->stream);
    for (size_t k = 0; k < n && e == hipSuccess; ++k) e = hipMemcpyAsync(&seen[k], base + k * CHUNK_BYTES, 8, hipMemcpyDeviceToHost, P->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(P->stream);
    *stale = false;
    for (size_t k = 0; k < n && e == hipSuccess; ++k) *stale = *stale || seen[k] != tags[k];
    return e;
}

}  // namespace

int kmers::pool_alloc(kmers_ctx *ctx, size_t bytes, int role, void **out) {
    *out = nullptr;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = attach(ctx, slot);
    if (!P) return KMERS_E_UNSUPPORTED;  // no virtual-memory management on this device: the caller falls back to hipMalloc
    Settle settle_on_return{P};          // (whatever this call unmaps is flushed before it returns)
    State &s = P->s;
    ++s.tick;
    // A lone output is written through two windows half an array apart: its MIDDLE is put on a chunk boundary (the block is the
    // two halves rounded up to whole chunks each, the caller's pointer lies `user_off` inside it), so that the first half is one
    // run of chunks and the second another, whatever the array's size (C3: 9.31 GiB -> 5 + 5 chunks, the pointer 0.34 GiB in).
    size_t user_off = 0, plan_bytes = bytes;
    const size_t n = block_chunks(bytes, role, &user_off, &plan_bytes);
    const Block *other = nullptr;
    const Block *partner = role == ROLE_DEFAULT ? partner_block(s, bytes, &other) : nullptr;
    // 1. a freed block of this shape that is still mapped: nothing to do but to come after the work that was queued when it was freed
    if (ctx->pool_cache > 0) {
        auto hit = find_cached(s, bytes, role, partner);
        if (hit != s.cached.end()) {
            const char *base = hit->first;
            wait_pending_on_stream(P, base, ctx->stream);
            Block b = from_cache(s, hit);
            b.req_bytes = bytes;
            b.user_off = user_off;
            b.serial = ++s.serial;
            s.blocks[base] = std::move(b);
            ++P->cache_hits;
            slot.generation.fetch_add(1, std::memory_order_release);
            *out = const_cast<char *>(base) + user_off;
            return KMERS_OK;
        }
    }
    ++P->cache_misses;
    age_out(P);  // (their chunks join the stock the plan draws on)
    {
        size_t free_b = 0, total_b = 0;  // a request the device could never hold must not walk all of its memory first
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && n * CHUNK_BYTES > free_b + s.held_bytes - s.in_use_bytes)
            return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: more than the device has free");
    }
    // 2. a new block, its chunks chosen by class.  The pool may grow past the request in search of the classes it wants: what it
    // walks past is held for the SEARCH only (trim_hoard below returns it to the driver), so the budget is generous -- up to
    // 128 GiB, at most half of what the device has free: one box begins with 70 GiB of ONE class in the order the driver hands
    // out handles, and a budget sized to a small request (16 GiB, tried first in round 6) left a 0.4 GB pair of arrays in one
    // class there.  A handle of memory the device has never used costs 3 us + 1 ms of probes.
    size_t search_limit = (size_t)128 << 30;
    if (ctx->pool_search_gib >= 0) {
        search_limit = (size_t)ctx->pool_search_gib << 30;
    } else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) search_limit = std::min(search_limit, free_b / 2);
    }
    std::vector<uint8_t> seq;
    float quality = 0.f;
    for (size_t searched = 0;;) {
        size_t free_counts[N_LISTS];
        for (int i = 0; i < N_LISTS; ++i) free_counts[i] = s.free_list[i].size();
        quality = 0.f;
        seq = plan(free_counts, plan_bytes, partner, role, &quality, other);
        if (!seq.empty() && (quality >= GOOD_PLAN || searched >= search_limit)) break;
        if (room_for_chunks(ctx, s, 1) && grow(P, ctx)) {
            if (!seq.empty()) searched += CHUNK_BYTES;
            continue;
        }
        if (evict_oldest(P)) continue;  // the device (or the cap) has no more: what the cache holds is stock too
        if (!seq.empty()) break;
        trim_hoard(P);
        return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: the device has no memory left for the pool");
    }
    std::vector<uint32_t> ids = take(s, seq);
    char *base = nullptr;
    uint32_t lead = 0;
    const bool home = n == 1;
    if (home) {
        base = s.chunks[ids[0]].home;  // one chunk: its own mapping is the block
    } else {
        if (!flush_tlb(P) || !reserve(&base, n * CHUNK_BYTES, &lead)) {
            give(s, ids);
            trim_hoard(P);
            return fail(ctx, KMERS_E_NOMEM, "kmers_dev_alloc: no address range (or no TLB flush) for a block of the pool");
        }
        hipError_t e = hipSuccess;
        size_t mapped = 0;
        for (; mapped < n && e == hipSuccess; ++mapped)
            e = hipMemMap(base + mapped * CHUNK_BYTES, CHUNK_BYTES, 0, static_cast<hipMemGenericAllocationHandle_t>(s.chunks[ids[mapped]].handle), 0);
        if (e != hipSuccess) --mapped;
        if (e == hipSuccess) e = hipMemSetAccess(base, n * CHUNK_BYTES, &P->access, 1);
        bool stale = false;
        if (e == hipSuccess) e = verify_block(P, base, ids, &stale);
        if (e == hipSuccess && stale) {  // flush and look again
            P->need_flush = true;
            if (flush_tlb(P)) e = verify_block(P, base, ids, &stale);
        }
        if (e != hipSuccess || stale) {
            for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap(base + i * CHUNK_BYTES, CHUNK_BYTES);
            unreserve(base, n * CHUNK_BYTES, lead);
            P->need_flush = true;
            give(s, ids);
            trim_hoard(P);
            if (e == hipSuccess) return fail(ctx, KMERS_E_HIP, "kmers_dev_alloc: a new block of the pool does not show the memory it was mapped to (stale translations)");
            return fail(ctx, KMERS_E_HIP, "kmers_dev_alloc: mapping a block of the pool", e);
        }
    }
    Block b;
    b.bytes = n * CHUNK_BYTES;
    b.req_bytes = bytes;
    b.user_off = user_off;
    b.va_lead = lead;
    b.chunks = std::move(ids);
    b.classes = std::move(seq);
    b.serial = ++s.serial;
    b.role = role;
    b.home = home;
    b.quality = quality;
    s.blocks[base] = std::move(b);
    trim_hoard(P);  // what the search walked past goes back to the driver
    slot.generation.fetch_add(1, std::memory_order_release);
    *out = base + user_off;
    return KMERS_OK;
}

int kmers::pool_free(kmers_ctx *ctx, void *p, bool *handled) {
    *handled = false;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = slot.pool;
    if (!P) return KMERS_OK;
    State &s = P->s;
    const char *base = nullptr;
    const Block *blk = block_of(s, p, 1, &base);
    if (!blk) {
        auto c = s.cached.upper_bound(static_cast<const char *>(p));
        if (c != s.cached.begin()) {
            --c;
            if (static_cast<const char *>(p) < c->first + c->second.bytes) {
                *handled = true;
                return fail(ctx, KMERS_E_BADARG, "kmers_dev_free: this block of the pool has been freed already");
            }
        }
        return KMERS_OK;
    }
    *handled = true;
    if (static_cast<const char *>(p) != base + blk->user_off) return fail(ctx, KMERS_E_BADARG, "kmers_dev_free: not the start of a block of the pool");
    Settle settle_on_return{P};
    attach(ctx, slot, false);  // (a context that only ever frees: its stream counts from now on)
    ++s.tick;
    auto it = s.blocks.find(base);
    Block b = std::move(it->second);
    s.blocks.erase(it);
    // No wait: whatever is queued on the streams of the device's contexts comes BEFORE the block's next use -- events recorded now,
    // waited for by the stream of whoever takes the block next (or by the host, if the block is taken apart first).
    record_pending(P, base);
    if (ctx->pool_cache > 0) {
        to_cache(s, base, std::move(b));
    } else {
        wait_pending_on_host(P, base);
        unmap_block(P, const_cast<char *>(base), b);
        give(s, b.chunks);
    }
    age_out(P);
    trim_hoard(P);
    slot.generation.fetch_add(1, std::memory_order_release);
    return KMERS_OK;
}

size_t kmers::pool_release_idle(kmers_ctx *ctx) {
    if (!ctx->slot) return 0;
    std::lock_guard<std::mutex> lock(ctx->slot->mu);
    kmers_device_pool *P = ctx->slot->pool;
    if (!P) return 0;
    const size_t before = P->s.held_bytes;
    release_idle_locked(P, true);
    settle(P);
    ctx->slot->generation.fetch_add(1, std::memory_order_release);
    return before - P->s.held_bytes;
}

void kmers::pool_detach(kmers_ctx *ctx) {
    if (!ctx->uses_pool) return;
    ctx->uses_pool = false;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = slot.pool;
    if (!P) return;
    P->users.erase(std::remove(P->users.begin(), P->users.end(), ctx), P->users.end());
    // kmers_ctx_destroy has waited for the context's stream, so what was recorded on it at the free of a block is complete -- and the
    // stream is about to go: an event that still names it must not reach hipStreamWaitEvent (the runtime looks at the stream an
    // event was recorded on; tests/test_gpu_pool.py, four threads: a segmentation fault in the thread that took a cached block
    // whose last user had closed its context).  Unless another context of the pool works on the same stream, they are dropped.
    bool shared = false;
    for (const kmers_ctx *u : P->users) shared = shared || u->stream == ctx->stream;
    if (!shared)
        for (auto it = P->pending.begin(); it != P->pending.end();) {
            auto &list = it->second;
            for (size_t i = 0; i < list.size();)
                if (list[i].stream == ctx->stream) {
                    (void)hipEventDestroy(list[i].ev);
                    list.erase(list.begin() + (long)i);
                } else {
                    ++i;
                }
            it = list.empty() ? P->pending.erase(it) : std::next(it);
        }
    (void)hipGetLastError();
    if (--P->refs > 0) return;
    // the last context of the device that used the pool: everything goes back, blocks that are still out included
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    while (evict_oldest(P)) {}
    for (auto &b : P->s.blocks) {
        unmap_block(P, const_cast<char *>(b.first), b.second);
        give(P->s, b.second.chunks);
    }
    P->s.blocks.clear();
    trim(P, true);
    P->need_flush = true;
    (void)flush_tlb(P);
    for (hipEvent_t e : P->spare_events) (void)hipEventDestroy(e);
    (void)hipEventDestroy(P->e0);
    (void)hipEventDestroy(P->e1);
    (void)hipStreamDestroy(P->stream);
    slot.pool = nullptr;
    slot.generation.fetch_add(1, std::memory_order_release);
    delete P;
}

// The launchers' two questions.  The answer stands for as long as no block of the pool has come or gone (slot.generation): a
// stream of launches into the same arrays takes no lock and walks no map.
float kmers::pool_arrays_differ(kmers_ctx *ctx, const void *a, size_t bytes_a, const void *b, size_t bytes_b) {
    if (!ctx->uses_pool) return -1.f;
    kmers_device_slot &slot = *ctx->slot;
    const uint64_t gen = slot.generation.load(std::memory_order_acquire);
    kmers_ctx::placement_answer &c = ctx->placed_pair;
    if (c.generation == gen && c.a == a && c.b == b && c.bytes_a == bytes_a && c.bytes_b == bytes_b) return c.differ;
    std::lock_guard<std::mutex> lock(slot.mu);
    c.generation = slot.generation.load(std::memory_order_acquire);
    c.a = a;
    c.b = b;
    c.bytes_a = bytes_a;
    c.bytes_b = bytes_b;
    c.differ = slot.pool ? arrays_differ(slot.pool->s, a, bytes_a, b, bytes_b) : -1.f;
    return c.differ;
}

float kmers::pool_halves_differ(kmers_ctx *ctx, const void *a, size_t bytes) {
    if (!ctx->uses_pool) return -1.f;
    kmers_device_slot &slot = *ctx->slot;
    const uint64_t gen = slot.generation.load(std::memory_order_acquire);
    kmers_ctx::placement_answer &c = ctx->placed_halves;
    if (c.generation == gen && c.a == a && c.bytes_a == bytes) return c.differ;
    std::lock_guard<std::mutex> lock(slot.mu);
    c.generation = slot.generation.load(std::memory_order_acquire);
    c.a = a;
    c.bytes_a = bytes;
    c.differ = slot.pool ? halves_differ(slot.pool->s, a, bytes) : -1.f;
    return c.differ;
}

extern "C" {

int kmers_pool_info(kmers_ctx *ctx, size_t *held, size_t *in_use, int *n_classes, size_t *class_bytes, double *two_class_gbps, double *one_class_gbps) {
    if (!ctx) return KMERS_E_BADARG;
    kmers_device_slot &slot = *ctx->slot;
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
    // GB/s of two store streams side by side, as the probes measured them (1 GiB each)
    const bool two = P && P->s.n_classes >= 2 && P->s.fast_ms > 0.f;
    if (two_class_gbps) *two_class_gbps = two ? 2.0 * (double)CHUNK_BYTES / 1e6 / (double)P->s.fast_ms : 0.0;
    if (one_class_gbps) *one_class_gbps = P && P->s.slow_ms > 0.f ? 2.0 * (double)CHUNK_BYTES / 1e6 / (double)P->s.slow_ms : 0.0;
    return KMERS_OK;
}

int kmers_pool_trim(kmers_ctx *ctx, size_t *released) {
    if (!ctx) return KMERS_E_BADARG;
    if (released) *released = 0;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.pool) return KMERS_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    kmers_device_pool *P = slot.pool;
    const size_t before = P->s.held_bytes;
    while (evict_oldest(P)) {}  // the cache of freed blocks first: their chunks join the free lists
    trim(P, P->s.blocks.empty());
    (void)flush_tlb(P);
    slot.generation.fetch_add(1, std::memory_order_release);
    if (released) *released = before - P->s.held_bytes;
    return KMERS_OK;
}

int kmers_pool_stats(kmers_ctx *ctx, uint64_t *out, size_t capacity) {
    if (!ctx || (!out && capacity)) return KMERS_E_BADARG;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    const kmers_device_pool *P = slot.pool;
    size_t idle = 0;
    if (P)
        for (const auto &l : P->s.free_list) idle += l.size() * CHUNK_BYTES;
    const uint64_t v[KMERS_POOL_STATS] = {P ? P->s.held_bytes : 0,   P ? P->s.in_use_bytes : 0, P ? P->s.cached_bytes : 0, idle,
                                          P ? P->cache_hits : 0,     P ? P->cache_misses : 0,   P ? P->evictions : 0,      P ? P->chunks_created : 0,
                                          P ? P->chunks_returned : 0, P ? P->n_probes : 0,      P ? P->s.cached.size() : 0, P ? P->s.blocks.size() : 0};
    for (size_t i = 0; i < capacity && i < KMERS_POOL_STATS; ++i) out[i] = v[i];
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
        hipLaunchKernelGGL(pool_probe_kernel, dim3((unsigned)groups), dim3(256), 0, ctx->stream, static_cast<ulonglong2 *>(a_dev),
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

int kmers_pool_layout(kmers_ctx *ctx, const void *block, size_t *chunk_bytes, unsigned char *classes, size_t capacity, size_t *n_chunks) {
    if (!ctx) return KMERS_E_BADARG;
    if (chunk_bytes) *chunk_bytes = CHUNK_BYTES;
    if (n_chunks) *n_chunks = 0;
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    if (!slot.pool) return KMERS_OK;
    const Block *b = block ? block_of(slot.pool->s, block, 1) : nullptr;
    if (!b) return KMERS_OK;
    if (n_chunks) *n_chunks = b->chunks.size();
    if (classes)
        for (size_t i = 0; i < b->classes.size() && i < capacity; ++i) classes[i] = b->classes[i];
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
    kmers_device_slot &slot = *ctx->slot;
    std::lock_guard<std::mutex> lock(slot.mu);
    kmers_device_pool *P = attach(ctx, slot);
    if (!P) return fail(ctx, KMERS_E_UNSUPPORTED, "no virtual-memory management on this device");
    State &s = P->s;
    if (!flush_tlb(P)) return fail(ctx, KMERS_E_NOMEM, "kmers_pool_selftest: the TLB flush could not run");  // (start clean: flushes are lazy)
    auto n_free = [&] {
        size_t total = 0;
        for (auto &l : s.free_list) total += l.size();
        return total;
    };
    while (n_free() < 2)
        if (!room_for_chunks(ctx, s, 1) || !grow(P, ctx)) return fail(ctx, KMERS_E_NOMEM, "kmers_pool_selftest: no memory");
    size_t free_counts[N_LISTS];
    for (int i = 0; i < N_LISTS; ++i) free_counts[i] = s.free_list[i].size();
    std::vector<uint32_t> ids = take(s, plan(free_counts, 2 * CHUNK_BYTES, nullptr, ROLE_DEFAULT, nullptr));
    if (ids.size() != 2) return fail(ctx, KMERS_E_NOMEM, "kmers_pool_selftest: two free chunks");
    int rc = KMERS_OK;
    for (int with_flush = 0; with_flush < 2 && rc == KMERS_OK; ++with_flush) {
        uint64_t tags[2] = {++P->tag, ++P->tag}, seen[2] = {0, 0};
        char *va[2] = {nullptr, nullptr};
        uint32_t lead[2] = {0, 0};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipMemcpy(s.chunks[ids[k]].home, &tags[k], 8, hipMemcpyHostToDevice);
            if (e == hipSuccess && !reserve(&va[k], CHUNK_BYTES, &lead[k])) e = hipErrorOutOfMemory;
            if (e != hipSuccess) break;
            e = hipMemMap(va[k], CHUNK_BYTES, 0, static_cast<hipMemGenericAllocationHandle_t>(s.chunks[ids[k]].handle), 0);
            if (e == hipSuccess) e = hipMemSetAccess(va[k], CHUNK_BYTES, &P->access, 1);
            if (e == hipSuccess) e = hipMemcpy(&seen[k], va[k], 8, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipMemUnmap(va[k], CHUNK_BYTES);
            if (e == hipSuccess) e = hipMemAddressFree(va[k] - lead[k], CHUNK_BYTES + VA_PAD);
            if (e == hipSuccess && with_flush) {
                P->need_flush = true;
                if (!flush_tlb(P)) e = hipErrorOutOfMemory;
            }
        }
        if (e != hipSuccess) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest", e);
        else if (seen[0] != tags[0]) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest: a fresh mapping does not show its chunk");
        else if (seen[1] != tags[1]) {
            if (with_flush) rc = fail(ctx, KMERS_E_HIP, "kmers_pool_selftest: stale translation in spite of the flush");
            else if (stale_without_flush) *stale_without_flush = 1;
        }
        P->need_flush = true;
        (void)flush_tlb(P);
    }
    give(s, ids);
    trim_hoard(P);
    (void)flush_tlb(P);
    return rc;
}

}  // extern "C"
