// class_pool.hpp -- the pure host logic of the device's CLASS POOL (pool_api.hip): which physical chunks make up a new block and
// in which order.  No HIP in here: tests/c/class_pool_check.cpp exercises it on the CPU.
//
// Why (profiles/r05_vmm.md): HBM on MI355X behaves as three REGION CLASSES of physical memory; store streams that run side by
// side inside one class share ~6.0-6.4 TB/s, streams in different classes reach ~7.1-7.2 TB/s (profiles/r03_alloc.md).  The class
// belongs to the PHYSICAL memory (tools/device_probes/vmm_va.hip: the same handles mapped in reverse order give the mirrored map, one handle
// mapped at many addresses a flat one).  Rounds 3-4 reserved most of HBM as one block and placed arrays inside its measured map;
// the pool instead takes physical memory from HIP's virtual-memory management in 1 GiB handles, measures the class of each
// once, and ASSEMBLES every block from handles of the classes it should have: the arrays of one launch in different classes at
// every relative position, the only output of a launch half in one class and half in another.  No reservation, no dependence on
// how fine-grained the physical map of a box is, arrays of any size.
//
// (Why not finer: hipMemMap takes whole handles, so a stripe is a handle.  Arrays striped A B A B at 2-64 MiB do reach the
// two-class rate with every write window, tools/device_probes/vmm_stripes.hip -- but the class of a 32 MiB handle cannot be measured on its own
// (a probe needs ~1 GiB per stream to tell 6.2 from 7.2 TB/s), and consecutive small handles are NOT neighbours in memory: the
// driver serves them from scattered fragments first, profiles/r05_pool_walk_32MiB_handles.txt.)
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

namespace kmers {
namespace pool {

constexpr size_t CHUNK_BYTES = (size_t)1 << 30;  // one physical handle, one probe stream
constexpr int MAX_CLASSES = 3;                   // what the device has (profiles/r05_vmm.md); a chunk that looks like none of them gets the nearest
constexpr uint8_t CLASS_UNKNOWN = MAX_CLASSES;   // the probes did not run: usable memory of no known class, its own free list
constexpr int N_LISTS = MAX_CLASSES + 1;
constexpr uint8_t NO_CLASS = 0xff;

constexpr int ROLE_DEFAULT = 0;      // == KMERS_ALLOC_DEFAULT: differ from the partner block (the most recent live one) at every relative position
constexpr int ROLE_LONE_OUTPUT = 1;  // == KMERS_ALLOC_LONE_OUTPUT: the second half differs from the first at every position (split-order launches)

struct Chunk {
    void *handle = nullptr;  // hipMemGenericAllocationHandle_t; nullptr: released
    char *home = nullptr;    // the chunk's own mapping, made once and kept: what the probes write through
    uint32_t va_lead = 0;    // bytes of the home RESERVATION in front of the mapping (pool_api.hip, reserve(): no mapping begins on a GiB boundary)
    uint8_t cls = CLASS_UNKNOWN;
    bool in_use = false;
    bool rep = false;  // the yardstick of its class: never handed out (a representative in use could not be probed against)
};
struct Block {
    size_t bytes = 0;      // of the reservation: chunks.size() * CHUNK_BYTES
    size_t req_bytes = 0;  // what was asked for: the array a launch writes
    size_t user_off = 0;   // where the caller's pointer lies inside the reservation (a lone output is shifted so that its MIDDLE is a chunk boundary)
    uint32_t va_lead = 0;  // bytes of the reservation in front of the block's base (pool_api.hip, reserve())
    std::vector<uint32_t> chunks;
    std::vector<uint8_t> classes;  // of the chunks, in order
    uint64_t serial = 0;
    int role = 0;          // ROLE_DEFAULT / ROLE_LONE_OUTPUT: what the block was assembled for
    bool home = false;     // ONE chunk, addressed through the chunk's own (home) mapping: no reservation of its own, nothing to map or unmap
    float quality = 1.f;   // of the plan it was assembled by (what a later request of its shape may settle for)
    uint64_t freed_tick = 0;  // cached blocks: State::tick when it was freed
};

struct State {
    std::vector<Chunk> chunks;
    std::vector<uint32_t> free_list[N_LISTS];  // per class; taken from the back
    std::map<const char *, Block> blocks;      // by base address: blocks that are OUT
    std::map<const char *, Block> cached;      // freed blocks that are still MAPPED (the next request of their shape takes one as it is)
    size_t cached_bytes = 0;
    uint64_t tick = 0;                         // counts allocations and frees: the age of a cached block
    uint32_t rep_chunk[MAX_CLASSES] = {};
    int n_classes = 0;
    float slow_ms = 0.f;       // two 1 GiB streams inside one class (the calibration's slowest pair)
    float fast_ms = 0.f;       // ... in two classes (the fastest probe so far; == slow_ms while only one class is known)
    size_t held_bytes = 0, in_use_bytes = 0;
    uint64_t serial = 0;
};

inline size_t chunks_for(size_t bytes) { return (bytes + CHUNK_BYTES - 1) / CHUNK_BYTES; }

// How many chunks of each class go to each SEGMENT of a new block so that as many positions as possible get a known class other
// than the one forbidden there: segments = maximal runs of positions with one forbidden class.  A transport problem with a handful
// of nodes on either side, solved exactly by augmenting paths (a segment that finds the classes it may take exhausted makes another
// segment switch to a class that still has stock), in three levels: known allowed classes; then chunks of unknown class; then the
// forbidden class itself for what is still missing.  A segment has a SECOND class to stay away from if it can (that of another
// recent block: with three classes the arrays of a launch and the sequence they are computed from all differ): given up first.
// take[s][c] = chunks of class c for segment s.
inline void assign_segments(const std::vector<uint8_t> &forbidden, const std::vector<uint8_t> &forbidden2, const std::vector<size_t> &seg_len,
                            const size_t free_counts[N_LISTS], std::vector<std::vector<size_t>> &take) {
    const size_t S = seg_len.size();
    take.assign(S, std::vector<size_t>(N_LISTS, 0));
    std::vector<size_t> need(seg_len), left(free_counts, free_counts + N_LISTS);
    for (int level = 0; level < 4; ++level) {
        auto allowed = [&](size_t s, int c) {
            if (c == forbidden[s]) return level >= 3;
            if (c == CLASS_UNKNOWN) return level >= 2;
            if (c == forbidden2[s]) return level >= 1;
            return true;
        };
        std::vector<size_t> order(S);  // longest segments first: they get the fullest classes, in one piece
        for (size_t s = 0; s < S; ++s) order[s] = s;
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return seg_len[a] > seg_len[b]; });
        for (size_t s : order) {
            while (need[s]) {
                int from_class[N_LISTS], found = -1;  // breadth-first over classes: -2 unvisited, -1 reached from s itself
                size_t from_seg[N_LISTS] = {};
                for (int c = 0; c < N_LISTS; ++c) from_class[c] = -2;
                std::vector<int> queue;
                {
                    std::vector<int> start;
                    for (int c = 0; c < N_LISTS; ++c)
                        if (allowed(s, c)) start.push_back(c);
                    std::stable_sort(start.begin(), start.end(), [&](int a, int b) { return left[a] > left[b]; });
                    for (int c : start) {
                        from_class[c] = -1;
                        queue.push_back(c);
                    }
                }
                for (size_t head = 0; head < queue.size() && found < 0; ++head) {
                    const int c = queue[head];
                    if (left[c]) {
                        found = c;
                        break;
                    }
                    for (size_t t = 0; t < S; ++t) {
                        if (t == s || !take[t][c]) continue;
                        for (int c2 = 0; c2 < N_LISTS; ++c2)
                            if (c2 != c && from_class[c2] == -2 && allowed(t, c2)) {
                                from_class[c2] = c;
                                from_seg[c2] = t;
                                queue.push_back(c2);
                            }
                    }
                }
                if (found < 0) break;
                size_t k = std::min(need[s], left[found]);
                for (int c = found; from_class[c] >= 0; c = from_class[c]) k = std::min(k, take[from_seg[c]][from_class[c]]);
                left[found] -= k;
                int c = found;
                for (; from_class[c] >= 0; c = from_class[c]) {
                    take[from_seg[c]][c] += k;
                    take[from_seg[c]][from_class[c]] -= k;
                }
                take[s][c] += k;
                need[s] -= k;
            }
        }
    }
}

// The classes of the chunks of a new block, in order, given the class each position should NOT have (NO_CLASS: any) and a second
// one it should avoid if it can.  Segments = maximal runs of one pair of forbidden classes; within a segment the chunks of one class stay together (long pure runs), and a run
// continues the class of the run before it when it can.  An empty result: fewer chunks are free than positions.
inline std::vector<uint8_t> plan_with(const size_t free_counts[N_LISTS], const std::vector<uint8_t> &forbidden_at,
                                      const std::vector<uint8_t> &second_at = std::vector<uint8_t>()) {
    const size_t n = forbidden_at.size();
    size_t total = 0;
    for (int i = 0; i < N_LISTS; ++i) total += free_counts[i];
    std::vector<uint8_t> seq;
    if (total < n || n == 0) return seq;
    std::vector<uint8_t> forbidden, forbidden2;
    std::vector<size_t> seg_len;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t f2 = second_at.empty() ? NO_CLASS : second_at[i];
        if (forbidden.empty() || forbidden.back() != forbidden_at[i] || forbidden2.back() != f2) {
            forbidden.push_back(forbidden_at[i]);
            forbidden2.push_back(f2);
            seg_len.push_back(0);
        }
        ++seg_len.back();
    }
    std::vector<std::vector<size_t>> take;
    assign_segments(forbidden, forbidden2, seg_len, free_counts, take);
    seq.reserve(n);
    for (size_t s = 0; s < seg_len.size(); ++s) {
        std::vector<int> cs;  // classes of the segment: the one the previous run ended with first, then by amount
        for (int c = 0; c < N_LISTS; ++c)
            if (take[s][c]) cs.push_back(c);
        std::sort(cs.begin(), cs.end(), [&](int a, int b) {
            const bool ca = !seq.empty() && seq.back() == a, cb = !seq.empty() && seq.back() == b;
            if (ca != cb) return ca;
            return take[s][a] > take[s][b];
        });
        for (int c : cs) seq.insert(seq.end(), take[s][c], (uint8_t)c);
    }
    return seq;
}

// fraction of the positions that got a class other than the forbidden one (both known); 1 where nothing is forbidden
inline float plan_quality(const std::vector<uint8_t> &seq, const std::vector<uint8_t> &forbidden_at) {
    if (seq.empty()) return 0.f;
    size_t ok = 0;
    for (size_t i = 0; i < seq.size(); ++i)
        ok += forbidden_at[i] == NO_CLASS || (seq[i] != forbidden_at[i] && seq[i] != CLASS_UNKNOWN && forbidden_at[i] != CLASS_UNKNOWN);
    return (float)ok / (float)seq.size();
}

// A new block of `bytes` bytes (n = chunks_for(bytes) chunks).
//   ROLE_DEFAULT: beside `partner` (the block it will most likely be written with; nullptr: none): chunk i should differ from the
//     partner's chunk at the same RELATIVE byte position of the two arrays (a launch writes element e of both at the same time),
//     and from `other`'s (another recent block: the sequence the arrays are computed from, a third array) if the stock allows.
//   ROLE_LONE_OUTPUT: the array is written through two windows half an array apart: the chunks of its second half should
//     differ from the chunks half an array before them.  The first half is taken from ONE class if one has the stock and leaves
//     enough of the others.
// *quality = the fraction of positions that got what they should; an empty result: fewer than n chunks are free.
inline std::vector<uint8_t> plan(const size_t free_counts[N_LISTS], size_t bytes, const Block *partner, int role, float *quality,
                                 const Block *other = nullptr) {
    const size_t n = chunks_for(bytes);
    std::vector<uint8_t> forbidden(n, NO_CLASS), seq;
    if (quality) *quality = 0.f;
    if (role == ROLE_LONE_OUTPUT && n >= 2) {
        const double half = (double)bytes / 2.0;
        size_t k1 = 0;  // chunks whose middle lies in the first half
        while (k1 < n && ((double)k1 + 0.5) * (double)CHUNK_BYTES < half) ++k1;
        k1 = std::min(std::max<size_t>(k1, 1), n - 1);
        size_t total = 0;
        for (int c = 0; c < N_LISTS; ++c) total += free_counts[c];
        if (total < n) return seq;
        std::vector<uint8_t> first;
        int pick = -1;
        for (int c = 0; c < MAX_CLASSES; ++c)
            if (free_counts[c] >= k1 && total - free_counts[c] >= n - k1 && (pick < 0 || free_counts[c] > free_counts[pick])) pick = c;
        if (pick >= 0) first.assign(k1, (uint8_t)pick);
        else first = plan_with(free_counts, std::vector<uint8_t>(k1, NO_CLASS));
        size_t left[N_LISTS];
        for (int c = 0; c < N_LISTS; ++c) left[c] = free_counts[c];
        for (uint8_t c : first) --left[c];
        std::vector<uint8_t> second_forbidden(n - k1);
        for (size_t j = k1; j < n; ++j) {  // the chunk half an array before the middle of chunk j
            const double at = ((double)j + 0.5) * (double)CHUNK_BYTES - half;
            second_forbidden[j - k1] = first[std::min(k1 - 1, (size_t)std::max(0.0, at / (double)CHUNK_BYTES))];
        }
        std::vector<uint8_t> second = plan_with(left, second_forbidden);
        seq = first;
        seq.insert(seq.end(), second.begin(), second.end());
        if (quality) *quality = plan_quality(second, second_forbidden);
        return seq;
    }
    std::vector<uint8_t> second(n, NO_CLASS);
    auto class_beside = [&](const Block *b, size_t i) -> uint8_t {
        if (!b || b->classes.empty() || !b->req_bytes) return NO_CLASS;
        const double x = std::min(1.0, ((double)i + 0.5) * (double)CHUNK_BYTES / (double)bytes);  // relative position of the chunk's middle
        return b->classes[std::min(b->classes.size() - 1, (size_t)(((double)b->user_off + x * (double)b->req_bytes) / (double)CHUNK_BYTES))];
    };
    for (size_t i = 0; i < n; ++i) {
        forbidden[i] = class_beside(partner, i);
        second[i] = class_beside(other, i);
    }
    seq = plan_with(free_counts, forbidden, second);
    if (quality) *quality = plan_quality(seq, forbidden);
    return seq;
}

// The same two questions for ARRAYS (what a launch writes: `bytes` from p on, inside a block), sampled at 64 relative positions.
inline uint8_t class_of_byte(const Block &b, const char *base, const char *p) { return b.classes[std::min(b.classes.size() - 1, (size_t)(p - base) / CHUNK_BYTES)]; }

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

inline float arrays_differ(const State &s, const void *pa, size_t bytes_a, const void *pb, size_t bytes_b) {
    const char *base_a = nullptr, *base_b = nullptr;
    const Block *a = pa && bytes_a ? block_of(s, pa, bytes_a, &base_a) : nullptr, *b = pb && bytes_b ? block_of(s, pb, bytes_b, &base_b) : nullptr;
    if (!a || !b) return -1.f;
    int ok = 0;
    for (int i = 0; i < 64; ++i) {
        const double x = (i + 0.5) / 64.0;
        const uint8_t ca = class_of_byte(*a, base_a, static_cast<const char *>(pa) + (size_t)(x * (double)bytes_a));
        const uint8_t cb = class_of_byte(*b, base_b, static_cast<const char *>(pb) + (size_t)(x * (double)bytes_b));
        ok += ca != cb && ca != CLASS_UNKNOWN && cb != CLASS_UNKNOWN;
    }
    return (float)ok / 64.f;
}
inline float halves_differ(const State &s, const void *p, size_t bytes) {
    const char *base = nullptr;
    const Block *b = p && bytes >= 2 ? block_of(s, p, bytes, &base) : nullptr;
    if (!b) return -1.f;
    int ok = 0;
    for (int i = 0; i < 64; ++i) {
        const double x = (i + 0.5) / 128.0;
        const uint8_t c0 = class_of_byte(*b, base, static_cast<const char *>(p) + (size_t)(x * (double)bytes));
        const uint8_t c1 = class_of_byte(*b, base, static_cast<const char *>(p) + (size_t)((x + 0.5) * (double)bytes));
        ok += c0 != c1 && c0 != CLASS_UNKNOWN && c1 != CLASS_UNKNOWN;
    }
    return (float)ok / 64.f;
}

// What a new block of `bytes` bytes will most likely be written beside: the most recent block still out that is at least half its
// size (the other array of the launch; a small block allocated in between -- the sequence, a scratch array -- must not take its
// place: with kmers, text, hashes allocated in this order the hashes landed in the kmers' class, 0.745 instead of 0.80).  *other:
// the most recent block apart from that one.
inline const Block *partner_block(const State &s, size_t bytes, const Block **other) {
    const Block *recent = nullptr, *second = nullptr, *partner = nullptr;
    for (const auto &b : s.blocks) {
        const Block *p = &b.second;
        if (!recent || p->serial > recent->serial) {
            second = recent;
            recent = p;
        } else if (!second || p->serial > second->serial) {
            second = p;
        }
        if (p->req_bytes >= bytes / 2 && (!partner || p->serial > partner->serial)) partner = p;
    }
    if (!partner) partner = recent;
    if (other) *other = recent != partner ? recent : second;
    return partner;
}

// ---- the cache of freed blocks, and how much the pool may hold beyond what is asked of it ---------------------------------------
// The reference's `collect` allocates a fresh Vector per call (Base.collect over src/iterators/CanonicalKmers.jl:199-225): a host
// loop is {allocate the outputs, launch, consume, free}.  Assembling a block costs a reservation, a map and a set-access per
// handle, a check of every handle and -- when an address range is re-used -- a TLB flush; freeing it used to cost a device-wide
// wait on top.  So a freed block stays MAPPED, and the next request of its shape (same number of chunks, same role, classes that
// suit the new partner as well as a fresh plan could) takes it as it is: no call into the driver at all.
constexpr uint64_t CACHE_AGE_TICKS = 32;       // a cached block nobody asked for over this many allocations / frees is taken apart
constexpr size_t HOARD_MIN_CHUNKS = 4;         // what the pool may hold beyond blocks (out or cached): representatives included,
constexpr size_t HOARD_FRACTION = 4;           //   max(HOARD_MIN_CHUNKS, 1 / HOARD_FRACTION of the chunks in blocks)
constexpr float GOOD_PLAN = 0.985f;            // the pool grows (within its search budget) until a block's plan is this good (0.95 until
                                               // round 6: an 80 GB array was accepted with 3 of its 75 handles in its partner's class, and the 1 Gbase
                                               // slice that lay on them ran 12 % slower than its neighbours, profiles/r06_n1.md)

// the classes a new block of `bytes` should NOT have, chunk by chunk, beside `partner` (plan()'s rule for ROLE_DEFAULT)
inline std::vector<uint8_t> forbidden_beside(const Block *partner, size_t bytes) {
    const size_t n = chunks_for(bytes);
    std::vector<uint8_t> forbidden(n, NO_CLASS);
    if (!partner || partner->classes.empty() || !partner->req_bytes) return forbidden;
    for (size_t i = 0; i < n; ++i) {
        const double x = std::min(1.0, ((double)i + 0.5) * (double)CHUNK_BYTES / (double)bytes);
        forbidden[i] = partner->classes[std::min(partner->classes.size() - 1, (size_t)(((double)partner->user_off + x * (double)partner->req_bytes) / (double)CHUNK_BYTES))];
    }
    return forbidden;
}

// how many chunks a request takes, and where the caller's pointer lies in them (a lone output of two chunks or more: its middle on
// a chunk boundary)
inline size_t block_chunks(size_t bytes, int role, size_t *user_off = nullptr, size_t *plan_bytes = nullptr) {
    size_t off = 0, pb = bytes;
    if (role == ROLE_LONE_OUTPUT && bytes >= 2 * CHUNK_BYTES) {
        const size_t half = (bytes / 2 + 4095) / 4096 * 4096, k1 = chunks_for(half);
        off = k1 * CHUNK_BYTES - half;
        pb = 2 * k1 * CHUNK_BYTES;
    }
    if (user_off) *user_off = off;
    if (plan_bytes) *plan_bytes = pb;
    return chunks_for(pb);
}

// The cached block a request should take, or cached.end(): the same number of chunks and the same role; for ROLE_DEFAULT its
// classes must differ from the partner's at GOOD_PLAN of the positions -- or at as many as the plan it was assembled by managed
// (a box whose stock cannot do better must not search again on every call).  The best one wins, the most recently freed on a tie.
inline std::map<const char *, Block>::iterator find_cached(State &s, size_t bytes, int role, const Block *partner) {
    const size_t n = block_chunks(bytes, role);
    auto best = s.cached.end();
    float best_q = -1.f;
    const std::vector<uint8_t> forbidden = role == ROLE_DEFAULT ? forbidden_beside(partner, bytes) : std::vector<uint8_t>();
    for (auto it = s.cached.begin(); it != s.cached.end(); ++it) {
        const Block &b = it->second;
        if (b.chunks.size() != n || b.role != role) continue;
        float q = 1.f;
        if (role == ROLE_DEFAULT) {
            q = plan_quality(b.classes, forbidden);
            if (q < std::min(GOOD_PLAN, b.quality)) continue;
        }
        if (q > best_q || (q == best_q && b.freed_tick > best->second.freed_tick)) {
            best = it;
            best_q = q;
        }
    }
    return best;
}

// chunks the pool holds outside blocks (free lists + representatives) beyond what it may: to be returned to the driver
inline size_t hoard_excess(const State &s) {
    size_t idle = (size_t)s.n_classes;  // the representatives
    for (const auto &l : s.free_list) idle += l.size();
    const size_t in_blocks = (s.in_use_bytes + s.cached_bytes) / CHUNK_BYTES;
    const size_t allowed = std::max(HOARD_MIN_CHUNKS, in_blocks / HOARD_FRACTION);
    return idle > allowed ? idle - allowed : 0;
}

// Take the chunks for the planned classes out of the free lists.
inline std::vector<uint32_t> take(State &s, const std::vector<uint8_t> &seq) {
    std::vector<uint32_t> out;
    out.reserve(seq.size());
    for (uint8_t c : seq) {
        const uint32_t id = s.free_list[c].back();
        s.free_list[c].pop_back();
        s.chunks[id].in_use = true;
        out.push_back(id);
    }
    s.in_use_bytes += out.size() * CHUNK_BYTES;
    return out;
}
inline void give(State &s, const std::vector<uint32_t> &chunks, bool from_cache = false) {
    for (size_t k = chunks.size(); k-- > 0;) {  // (back to front: the next block of this shape gets them in the same order)
        Chunk &c = s.chunks[chunks[k]];
        c.in_use = false;
        s.free_list[c.cls].push_back(chunks[k]);
    }
    (from_cache ? s.cached_bytes : s.in_use_bytes) -= chunks.size() * CHUNK_BYTES;
}
// a block that is out goes to the cache as it is / a cached block goes out again
inline void to_cache(State &s, const char *base, Block &&b) {
    b.freed_tick = s.tick;
    s.in_use_bytes -= b.bytes;
    s.cached_bytes += b.bytes;
    s.cached[base] = std::move(b);
}
inline Block from_cache(State &s, std::map<const char *, Block>::iterator it) {
    Block b = std::move(it->second);
    s.cached.erase(it);
    s.cached_bytes -= b.bytes;
    s.in_use_bytes += b.bytes;
    return b;
}

}  // namespace pool
}  // namespace kmers
