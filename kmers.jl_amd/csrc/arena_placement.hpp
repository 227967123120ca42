// arena_placement.hpp -- the pure host logic of the context's device-memory arena (memory_api.hip): the block's measured region
// map and where a new block goes.  No HIP in here: tests/c/arena_placement_check.cpp exercises it on the CPU with made-up maps.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <map>
#include <vector>

constexpr size_t KMERS_ARENA_GRANULE_BYTES = (size_t)2 << 20;   // == KMERS_ARENA_GRANULE of include/kmers_hip.h
constexpr size_t KMERS_ARENA_REGION_BYTES = (size_t)4 << 30;    // granule of the region map

// The context's device-memory arena (kmers_arena_reserve, include/kmers_hip.h): ONE hipMalloc, sub-allocated in 2 MiB
// granules by kmers_dev_alloc.  Offsets are relative to `base`; free ranges are kept coalesced.  `region` is the map of the
// block that the calibration of memory_api.hip measured: two store streams inside one REGION CLASS of HBM share a write rate of
// ~6 TB/s on MI355X, streams in different classes reach ~7.1 TB/s, so a new block goes where it writes fastest beside the live ones.
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
    float one_class_rate = 0.f;            // measured: GB/s of two store streams inside ONE granule (the median over the granules)
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
// (arrays of different lengths -- two-word kmers and their one-word hashes -- are compared at the same RELATIVE places: that is
// where a launch writes them at the same time)
inline bool kmers_arena_spread(const kmers_arena &a, const void *p, size_t bytes_p, const void *q, size_t bytes_q) {
    if (a.run_start.empty() || !p || !q || bytes_p == 0 || bytes_q == 0) return false;
    const char *cp = static_cast<const char *>(p), *cq = static_cast<const char *>(q);
    if (cp < a.base || cp + bytes_p > a.base + a.bytes || cq < a.base || cq + bytes_q > a.base + a.bytes) return false;
    const size_t k = a.run_start.size();
    float sum = 0.f;
    for (int i = 0; i < 8; ++i) {
        const size_t tp = (size_t)((2 * i + 1) * (double)bytes_p / 16.0), tq = (size_t)((2 * i + 1) * (double)bytes_q / 16.0);
        sum += a.pair_rate[kmers_arena_run_of(a, (size_t)(cp - a.base) + tp) * k + kmers_arena_run_of(a, (size_t)(cq - a.base) + tq)];
    }
    return sum / 8.f >= 0.95f * a.best_pair_rate;
}
inline bool kmers_arena_spread(const kmers_arena &a, const void *p, const void *q, size_t bytes) {
    return kmers_arena_spread(a, p, bytes, q, bytes);
}

// true iff the array of `bytes` bytes at p lies in the arena ACROSS a class boundary, so that its two halves -- the two write
// windows of a split-order launch (stream_kernel.hpp: even workgroups walk the first half of the tiles, odd ones the second)
// -- are written side by side at a measured rate within 5 % of the block's best pair (four sample points per half).  A launch
// with ONE output array is then worth writing through two windows (profiles/r03_tuning.md section 5).
inline bool kmers_arena_straddles(const kmers_arena &a, const void *p, size_t bytes) {
    if (a.run_start.size() < 2 || !p || bytes < 16) return false;
    const char *cp = static_cast<const char *>(p);
    if (cp < a.base || cp + bytes > a.base + a.bytes) return false;
    const size_t k = a.run_start.size(), off = (size_t)(cp - a.base), half = bytes / 2;
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) {
        const size_t t = (size_t)((2 * i + 1) * (double)half / 8.0);
        sum += a.pair_rate[kmers_arena_run_of(a, off + t) * k + kmers_arena_run_of(a, off + half + t)];
    }
    return sum / 4.f >= 0.95f * a.best_pair_rate;
}

namespace kmers {
namespace arena {

constexpr size_t GRANULE = KMERS_ARENA_GRANULE_BYTES;
constexpr size_t REGION = KMERS_ARENA_REGION_BYTES;
inline size_t round_up(size_t x) { return (x + GRANULE - 1) / GRANULE * GRANULE; }

inline size_t run_of(const kmers_arena &a, size_t off) { return kmers_arena_run_of(a, off); }
inline int class_at(const kmers_arena &a, size_t off) { return a.run_class.empty() ? 0 : a.run_class[run_of(a, off)]; }
inline size_t run_end(const kmers_arena &a, size_t i) { return i + 1 < a.run_start.size() ? a.run_start[i + 1] : a.bytes; }

inline void arena_commit(kmers_arena &a, std::map<size_t, size_t>::iterator range, size_t off, size_t need) {
    const size_t fo = range->first, fl = range->second;
    a.free_ranges.erase(range);
    if (off > fo) a.free_ranges[fo] = off - fo;
    if (fo + fl > off + need) a.free_ranges[off + need] = fo + fl - (off + need);
    a.used[off] = need;
    a.last2_run = a.last_run;
    a.last_run = a.run_start.empty() ? -1 : (int)run_of(a, off);
    a.last_off = off;
    a.last_len = need;
}

// Placement.  With a region map every stretch of a free range inside ONE run that fits the request is a candidate; the run
// whose MEASURED two-stream rate beside the runs of the live blocks is highest on average wins (the previous allocation counts
// double); ties go to the tightest stretch.  Without a map,
// or when no stretch fits (a request larger than any run): best fit over the free ranges.
// (Centring EVERY large block on a class boundary and writing every array through two windows costs a two-output launch 1-4 %
// against its two arrays in two different classes, profiles/r03_alloc.md; the array that is the ONLY output of its launches is
// the one that gains -- arena_take_straddling below, asked for by role.)
inline bool arena_take(kmers_arena &a, size_t need, size_t *off_out) {
    // A block of a quarter of the arena or more (the 80 GB arrays of a 10 Gbase launch) is not put inside one run even where one
    // would hold it: the FIRST goes to the bottom of the block, which leaves the next one the whole rest to choose from, and the
    // next is placed by the sampled search below (on a box with the map A16 B24 C13 the first array inside B left the second
    // 71 % of its length beside another class; bottom + search: all of it).
    const bool huge = need >= a.bytes / 4;
    if (!a.run_start.empty() && !huge) {
        auto best_range = a.free_ranges.end();
        size_t best_off = 0, best_slack = 0;
        float best_score = -1.f;
        const size_t k = a.run_start.size();
        for (auto it = a.free_ranges.begin(); it != a.free_ranges.end(); ++it) {
            const size_t fo = it->first, fe = fo + it->second;
            size_t pos = fo;
            while (pos < fe) {  // the run that holds `pos`, cut to the free range
                const size_t r = run_of(a, pos);
                const size_t stretch_end = std::min(fe, run_end(a, r));
                if (stretch_end - pos >= need) {
                    // mean GB/s beside the LIVE blocks of the arena (those of 64 MiB or more: the arrays and sequences launches
                    // stream through; the previous allocation counts double -- the arrays of one launch are allocated one after
                    // the other); nothing live: the run's own rate
                    float sum = 0.f, weight = 0.f;
                    for (const auto &u : a.used) {
                        if (u.second < ((size_t)64 << 20)) continue;
                        const float w = u.first == a.last_off ? 2.f : 1.f;
                        // (a block that lies across a boundary -- a lone output -- counts with the runs of both its ends)
                        sum += w * 0.5f * (a.pair_rate[run_of(a, u.first) * k + r] + a.pair_rate[run_of(a, u.first + u.second - 1) * k + r]);
                        weight += w;
                    }
                    float score = weight > 0.f ? sum / weight : a.pair_rate[r * k + r];
                    score = (float)(int)(score / 100.f);  // (rates within 100 GB/s of each other are a tie)
                    const size_t slack = stretch_end - pos - need;
                    if (score > best_score || (score == best_score && slack < best_slack)) {
                        best_range = it;
                        best_off = pos;
                        best_slack = slack;
                        best_score = score;
                    }
                }
                pos = stretch_end;
            }
        }
        if (best_range != a.free_ranges.end()) {
            arena_commit(a, best_range, best_off, need);
            *off_out = best_off;
            return true;
        }
    }
    if (!a.run_start.empty() && a.last_len) {
        // a block longer than any run (the 80 GB arrays of a 10 Gbase launch) passes through several runs: wherever in a free
        // range that fits its runs -- sampled at sixteen points -- write fastest beside the previous block's.  The two ends of
        // the range come first (nothing is fragmented); a position inside it, tried every 2 GiB, must be better than the better
        // end by 100 GB/s on average (a fine-grained map -- runs of 16-32 GiB -- leaves the ends a matter of luck: the 10 Gbase
        // launch ran at 0.82 on such a box and at 0.88-0.89 on boxes with runs of 64 GiB).
        const size_t k = a.run_start.size();
        auto best_range = a.free_ranges.end();
        size_t best_off = 0;
        float best_score = -1.f;
        auto score_at = [&](size_t off) {
            float sum = 0.f;
            for (int i = 0; i < 16; ++i) {
                const size_t t = (size_t)((2 * i + 1) * (double)need / 32.0), u = (size_t)((2 * i + 1) * (double)a.last_len / 32.0);
                sum += a.pair_rate[run_of(a, a.last_off + u) * k + run_of(a, off + t)];
            }
            return sum / 16.f;
        };
        for (auto it = a.free_ranges.begin(); it != a.free_ranges.end(); ++it) {
            if (it->second < need) continue;
            const size_t lo = it->first, hi = (it->first + it->second - need) / GRANULE * GRANULE;
            for (int e = 0; e < 2; ++e) {  // the ends: a later one must be better by 50 GB/s
                const size_t off = e ? hi : lo;
                if (off < lo) continue;
                const float sc = score_at(off);
                if (sc > best_score + 50.f) {
                    best_score = sc;
                    best_range = it;
                    best_off = off;
                }
            }
        }
        const float end_score = best_score;
        // (a place inside a range splits it: only when no further block of this size would fit into the arena anyway)
        size_t free_total = 0;
        for (const auto &f : a.free_ranges) free_total += f.second;
        const bool may_split = free_total - need < need;
        for (auto it = a.free_ranges.begin(); may_split && it != a.free_ranges.end(); ++it) {
            if (it->second < need) continue;
            const size_t lo = it->first, hi = (it->first + it->second - need) / GRANULE * GRANULE;
            for (size_t off = lo + REGION / 2; off < hi; off += REGION / 2) {
                const float sc = score_at(off);
                if (sc > end_score + 100.f && sc > best_score) {
                    best_score = sc;
                    best_range = it;
                    best_off = off;
                }
            }
        }
        if (best_range != a.free_ranges.end()) {
            arena_commit(a, best_range, best_off, need);
            *off_out = best_off;
            return true;
        }
    }
    auto best = a.free_ranges.end();
    for (auto it = a.free_ranges.begin(); it != a.free_ranges.end(); ++it)
        if (it->second >= need && (best == a.free_ranges.end() || it->second < best->second)) best = it;
    if (best == a.free_ranges.end()) return false;
    const size_t off = best->first;
    arena_commit(a, best, off, need);
    *off_out = off;
    return true;
}

// Placement of a LONE output (KMERS_ALLOC_LONE_OUTPUT: the only array the launches that fill it write): centred on the
// boundary between two runs of different classes, so that a split-order launch writes its two halves into two classes at once
// (C3 0.81 -> 0.87 of 8 TB/s, profiles/r03_tuning.md section 5).  The boundary whose two runs write fastest side by side wins;
// one whose runs are shorter than half the block only if no other fits.  false: no boundary with room around it (or no map).
inline bool arena_take_straddling(kmers_arena &a, size_t need, size_t *off_out) {
    const size_t k = a.run_start.size();
    if (k < 2) return false;
    auto best_range = a.free_ranges.end();
    size_t best_off = 0;
    float best_score = -1.f;
    for (size_t r = 0; r + 1 < k; ++r) {
        const size_t b = a.run_start[r + 1];
        if (b < need / 2) continue;
        const size_t off = (b - need / 2) / GRANULE * GRANULE;
        if (off + need > a.bytes) continue;
        auto it = a.free_ranges.upper_bound(off);
        if (it == a.free_ranges.begin()) continue;
        --it;
        if (it->first > off || it->first + it->second < off + need) continue;  // not free
        float score = a.pair_rate[r * k + (r + 1)];
        if (need / 2 > b - a.run_start[r] || need / 2 > run_end(a, r + 1) - b) score -= 1000.f;  // a half reaches into a third run
        if (score > best_score) {
            best_score = score;
            best_range = it;
            best_off = off;
        }
    }
    if (best_range == a.free_ranges.end()) return false;
    arena_commit(a, best_range, best_off, need);
    *off_out = best_off;
    return true;
}

inline void arena_give(kmers_arena &a, size_t off, size_t len) {
    if (off == a.last_off) a.last_len = 0;  // the "previous allocation" the placement scores against is gone
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

}  // namespace arena
}  // namespace kmers
