// stripe_pool.hpp -- the pure host logic of the device's STRIPED POOL (pool_api.hip): which physical chunks make up a new block,
// in which order, and when the pool has to grow for it.  No HIP in here: tests/c/stripe_pool_check.cpp exercises it on the CPU.
//
// Why (profiles/r05_vmm.md): HBM on MI355X behaves as three REGION CLASSES of physical memory; store streams that run side by
// side inside one class share ~6.0-6.4 TB/s, streams in different classes reach ~7.1-7.2 TB/s (profiles/r03_alloc.md).  The class
// belongs to the PHYSICAL memory (the same handles mapped in reverse order give the mirrored map; one handle mapped at many
// addresses gives a flat one), so an array assembled with HIP's virtual-memory management from chunks of ALTERNATING classes has
// both classes inside the write window of any launch: ONE array written 16 KiB per workgroup 7.05-7.18 TB/s at stripes of
// 2-64 MiB against 6.0-6.45 inside one class, two arrays 7.15-7.24 in either phase.  Placement becomes a property of every
// array, not of where a 230 GB reservation happens to lie.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

namespace kmers {
namespace pool {

constexpr size_t CHUNK_BYTES = (size_t)32 << 20;  // one physical handle = one stripe (hipMemMap takes whole handles only)
constexpr uint32_t UNIT_CHUNKS = 32;              // chunks created together and classified together: 1 GiB, two halves of 512 MiB
constexpr size_t UNIT_BYTES = CHUNK_BYTES * UNIT_CHUNKS;
constexpr int MAX_CLASSES = 4;                    // three have been seen; a fourth label absorbs a noisy probe
constexpr uint8_t CLASS_UNKNOWN = MAX_CLASSES;    // no representative was free to compare with: its own free list
constexpr int N_LISTS = MAX_CLASSES + 1;

struct Chunk {
    void *handle = nullptr;  // hipMemGenericAllocationHandle_t
    uint32_t unit = 0;
    uint8_t cls = CLASS_UNKNOWN;
    bool in_use = false;
    bool rep = false;  // part of a class representative: never handed out (a representative in use cannot be probed against, and
                       // units that cannot be compared with every class stay unclassified) -- 512 MiB per class is the pool's yardstick
};
struct Unit {
    char *home = nullptr;  // the unit's own mapping, made once and kept: what the probes write through
    uint32_t first_chunk = 0;
    uint32_t in_use = 0;
    bool released = false;
};
struct Block {
    size_t bytes = 0;  // of the reservation: chunks.size() * CHUNK_BYTES
    std::vector<uint32_t> chunks;
    float alternation = 0.f;  // fraction of neighbouring chunk pairs of different (known) classes
};

struct State {
    std::vector<Chunk> chunks;
    std::vector<Unit> units;
    std::vector<uint32_t> free_list[N_LISTS];  // per class; taken from the back (representatives are not in them)
    std::map<const char *, Block> blocks;      // by base address
    char *rep_ptr[MAX_CLASSES] = {};           // 512 MiB of each class (half a unit, through the unit's home mapping)
    uint32_t rep_unit[MAX_CLASSES] = {};
    uint32_t rep_half[MAX_CLASSES] = {};
    int n_classes = 0;
    float slow_ms = 0.f;       // the slowest probe so far: two streams inside one class
    float fast_ms = 0.f;       // the fastest: two classes
    size_t held_bytes = 0, in_use_bytes = 0;
};

inline size_t chunks_for(size_t bytes) { return (bytes + CHUNK_BYTES - 1) / CHUNK_BYTES; }

// How many chunks of each free list a block of n chunks takes: as EVEN as the lists allow (water-filling: the smallest level m
// with sum min(free_i, m) >= n; what is over at the level goes to the longest lists).  false: fewer than n chunks are free.
inline bool pick_counts(const size_t free_counts[N_LISTS], size_t n, size_t counts[N_LISTS]) {
    size_t total = 0;
    for (int i = 0; i < N_LISTS; ++i) total += free_counts[i];
    for (int i = 0; i < N_LISTS; ++i) counts[i] = 0;
    if (total < n) return false;
    size_t lo = 0, hi = n;  // smallest level that reaches n
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        size_t s = 0;
        for (int i = 0; i < N_LISTS; ++i) s += std::min(free_counts[i], mid);
        if (s >= n) hi = mid;
        else lo = mid + 1;
    }
    size_t s = 0;
    for (int i = 0; i < N_LISTS; ++i) {
        counts[i] = std::min(free_counts[i], lo);
        s += counts[i];
    }
    // the level overshoots by less than the number of lists: give back from the lists that sit AT the level, shortest first
    for (int pass = 0; s > n && pass < N_LISTS; ++pass) {
        int best = -1;
        for (int i = 0; i < N_LISTS; ++i)
            if (counts[i] == lo && counts[i] > 0 && (best < 0 || free_counts[i] < free_counts[best])) best = i;
        if (best < 0) break;
        --counts[best];
        --s;
    }
    return s == n;
}

// true iff a block of n chunks can be put together without two neighbours of one class: no list has to give more than half
// (rounded up).  The unknown list counts as a class of its own (its chunks may or may not differ from their neighbours).
inline bool balanced(const size_t free_counts[N_LISTS], size_t n) {
    size_t counts[N_LISTS];
    if (!pick_counts(free_counts, n, counts)) return false;
    if (n < 2) return true;
    for (int i = 0; i < N_LISTS; ++i)
        if (counts[i] > (n + 1) / 2) return false;
    return true;
}

// The order of the classes along the block: always the class with the most chunks left that is not the previous one (the
// classic rearrangement: neighbours differ whenever no class holds more than half).  Ties go round-robin (A B C A B C).
inline std::vector<uint8_t> stripe_order(const size_t counts_in[N_LISTS], size_t n) {
    size_t left[N_LISTS];
    for (int i = 0; i < N_LISTS; ++i) left[i] = counts_in[i];
    std::vector<uint8_t> order;
    order.reserve(n);
    int prev = -1;
    for (size_t k = 0; k < n; ++k) {
        int best = -1;
        for (int step = 1; step <= N_LISTS; ++step) {  // start behind the previous class: round-robin among equals
            const int i = (prev + step + N_LISTS) % N_LISTS;
            if (i == prev || left[i] == 0) continue;
            if (best < 0 || left[i] > left[best]) best = i;
        }
        if (best < 0) best = prev;  // only the previous class is left
        if (best < 0 || left[best] == 0) break;
        order.push_back((uint8_t)best);
        --left[best];
        prev = best;
    }
    return order;
}

inline float alternation_of(const State &s, const std::vector<uint32_t> &chunks) {
    if (chunks.size() < 2) return 0.f;
    size_t differ = 0;
    for (size_t i = 1; i < chunks.size(); ++i) {
        const uint8_t a = s.chunks[chunks[i - 1]].cls, b = s.chunks[chunks[i]].cls;
        differ += a != b && a != CLASS_UNKNOWN && b != CLASS_UNKNOWN;
    }
    return (float)differ / (float)(chunks.size() - 1);
}

// Take the chunks of a new block out of the free lists (the caller has made sure n are free).  Within a class the chunks come
// in the order the lists hold them: most recently freed / created first.
inline std::vector<uint32_t> take(State &s, size_t n) {
    size_t free_counts[N_LISTS], counts[N_LISTS];
    for (int i = 0; i < N_LISTS; ++i) free_counts[i] = s.free_list[i].size();
    std::vector<uint32_t> out;
    if (!pick_counts(free_counts, n, counts)) return out;
    const std::vector<uint8_t> order = stripe_order(counts, n);
    out.reserve(n);
    for (uint8_t c : order) {
        const uint32_t id = s.free_list[c].back();
        s.free_list[c].pop_back();
        s.chunks[id].in_use = true;
        ++s.units[s.chunks[id].unit].in_use;
        out.push_back(id);
    }
    s.in_use_bytes += out.size() * CHUNK_BYTES;
    return out;
}

inline void give(State &s, const std::vector<uint32_t> &chunks) {
    for (size_t k = chunks.size(); k-- > 0;) {  // (back to front: the next block of this size gets them in the same order)
        const uint32_t id = chunks[k];
        Chunk &c = s.chunks[id];
        c.in_use = false;
        --s.units[c.unit].in_use;
        s.free_list[c.cls].push_back(id);
    }
    s.in_use_bytes -= chunks.size() * CHUNK_BYTES;
}

// the block that holds [p, p + bytes), or nullptr
inline const Block *block_of(const State &s, const void *p, size_t bytes, const char **base_out = nullptr) {
    const char *c = static_cast<const char *>(p);
    auto it = s.blocks.upper_bound(c);
    if (it == s.blocks.begin()) return nullptr;
    --it;
    if (c < it->first || c + bytes > it->first + it->second.bytes) return nullptr;
    if (base_out) *base_out = it->first;
    return &it->second;
}

}  // namespace pool
}  // namespace kmers
