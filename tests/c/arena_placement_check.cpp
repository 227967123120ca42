// The pure host logic of the context's arena (kmers.jl_amd/csrc/arena_placement.hpp) on made-up region maps: best fit and
// merging without a map; with a map, the blocks of a launch in different classes, the sequence in a third one, blocks longer
// than a run at the ends of a free range, kmers_arena_spread; and the allocator's invariants under a random sequence of requests.
// Plain C++ (tests/test_capi_abi.py compiles and runs it with g++; no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <random>
#include <set>

#include "../../kmers.jl_amd/csrc/arena_placement.hpp"

using namespace kmers::arena;

static int failures = 0;
#define EXPECT(cond)                                                     \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::printf("FAILED line %d: %s\n", __LINE__, #cond);        \
            ++failures;                                                  \
        }                                                                \
    } while (0)

constexpr size_t GiB = (size_t)1 << 30, MiB = (size_t)1 << 20;

static kmers_arena fresh(size_t bytes) {
    kmers_arena a;
    static char fake_base;  // (addresses are never dereferenced)
    a.base = &fake_base;
    a.bytes = bytes;
    a.free_ranges[0] = bytes;
    return a;
}

// runs [start GiB, class] with 6000 GB/s inside a class and 7000 between classes
static void set_map(kmers_arena &a, std::vector<std::pair<size_t, int>> runs) {
    const size_t k = runs.size();
    for (auto &r : runs) {
        a.run_start.push_back(r.first * GiB);
        a.run_class.push_back((uint8_t)r.second);
    }
    a.pair_rate.assign(k * k, 0.f);
    for (size_t i = 0; i < k; ++i)
        for (size_t j = 0; j < k; ++j) a.pair_rate[i * k + j] = runs[i].second == runs[j].second ? 6000.f : 7000.f;
    a.best_pair_rate = 7000.f;
    a.region_bytes = REGION;
    a.n_classes = 3;
}

static bool take(kmers_arena &a, size_t bytes, size_t *off) { return arena_take(a, round_up(bytes), off); }
static void give(kmers_arena &a, size_t off) {
    auto it = a.used.find(off);
    const size_t len = it->second;
    a.used.erase(it);
    arena_give(a, off, len);
}

int main() {
    {  // ---- no map: best fit from the bottom, merging
        kmers_arena a = fresh(130 * MiB);
        size_t x, y, z, w;
        EXPECT(take(a, 6 * MiB - 7, &x) && x == 0);
        EXPECT(take(a, 20 * MiB, &y) && y == 6 * MiB);
        EXPECT(take(a, 1, &z) && z == 26 * MiB);
        give(a, y);
        EXPECT(take(a, 18 * MiB, &w) && w == y);          // the 20 MiB hole, not the tail
        size_t v;
        EXPECT(take(a, 2 * MiB, &v) && v == y + 18 * MiB);  // what is left of the hole
        size_t big;
        EXPECT(!take(a, 120 * MiB, &big));                  // the caller falls through to a plain allocation
        give(a, x); give(a, z); give(a, w); give(a, v);
        EXPECT(a.used.empty() && a.free_ranges.size() == 1 && a.free_ranges.begin()->first == 0 && a.free_ranges.begin()->second == 130 * MiB);
    }
    {  // ---- the finest map a box of round 4 showed (4 GiB granules: A16 B7 C1 A1 B3 C1 A1 B3 C3 A1 C4 A1 B4 C5 A1 B1,
       //      profiles/r04_shape.md): the two 8 GB outputs of the headline launch still land each inside ONE run, in two classes
        kmers_arena a = fresh(212 * GiB);
        std::vector<std::pair<size_t, int>> runs;
        const int cls[] = {0, 1, 2, 0, 1, 2, 0, 1, 2, 0, 2, 0, 1, 2, 0, 1};
        const size_t len[] = {16, 7, 1, 1, 3, 1, 1, 3, 3, 1, 4, 1, 4, 5, 1, 1};
        size_t at = 0;
        for (int i = 0; i < 16; ++i) {
            runs.push_back({at, cls[i]});
            at += 4 * len[i];
        }
        EXPECT(at == 212);
        set_map(a, runs);
        size_t k1, k2, src;
        EXPECT(take(a, (size_t)8e9, &k1));
        EXPECT(take(a, (size_t)8e9, &k2));
        EXPECT(take(a, (size_t)5e8, &src));
        EXPECT(run_of(a, k1) == run_of(a, k1 + (size_t)8e9 - 1) && run_of(a, k2) == run_of(a, k2 + (size_t)8e9 - 1));
        EXPECT(class_at(a, k1) != class_at(a, k2));
        EXPECT(kmers_arena_spread(a, a.base + k1, a.base + k2, (size_t)8e9));
        give(a, k1); give(a, k2); give(a, src);
    }
    {  // ---- a map: A [0,64) B [64,128) C [128,160) A [160,200) GiB
        kmers_arena a = fresh(200 * GiB);
        set_map(a, {{0, 0}, {64, 1}, {128, 2}, {160, 0}});
        size_t k1, k2, src, tiny;
        EXPECT(take(a, 8 * GiB, &k1));
        EXPECT(take(a, 8 * GiB, &k2));
        EXPECT(class_at(a, k1) != class_at(a, k2));                                    // the two outputs of a launch
        EXPECT(class_at(a, k1) == class_at(a, k1 + 8 * GiB - 1) && class_at(a, k2) == class_at(a, k2 + 8 * GiB - 1));  // each inside one run
        EXPECT(take(a, GiB / 2, &src));
        EXPECT(class_at(a, src) != class_at(a, k1) && class_at(a, src) != class_at(a, k2));  // the sequence: the third class
        EXPECT(take(a, MiB, &tiny));                                                    // small blocks are placed, not steered by
        EXPECT(kmers_arena_spread(a, a.base + k1, a.base + k2, 8 * GiB));
        EXPECT(!kmers_arena_spread(a, a.base + k1, a.base + k1 + 4 * GiB, 4 * GiB));    // the two halves of one block: one class
        EXPECT(!kmers_arena_spread(a, a.base + k1, a.base + 300 * GiB, GiB));           // not in the arena
        EXPECT(!kmers_arena_spread(a, nullptr, a.base + k2, GiB));
        give(a, k1); give(a, k2); give(a, src); give(a, tiny);
        EXPECT(a.used.empty() && a.free_ranges.size() == 1 && a.free_ranges.begin()->second == 200 * GiB);
        // blocks longer than any run: both must fit, each at an end of a free range
        size_t b1, b2, s2;
        EXPECT(take(a, 75 * GiB, &b1) && (b1 == 0 || b1 + round_up(75 * GiB) == 200 * GiB));
        EXPECT(take(a, 75 * GiB, &b2));
        EXPECT(b2 >= b1 + 75 * GiB || b2 + 75 * GiB <= b1);
        EXPECT(take(a, 5 * GiB, &s2));
        EXPECT(a.used.size() == 3);
        give(a, b1); give(a, b2); give(a, s2);
        EXPECT(a.free_ranges.size() == 1);
    }
    {  // ---- the lone output of a launch: across a class boundary, halves inside the two runs; detection by the launcher's query
        kmers_arena a = fresh(200 * GiB);
        set_map(a, {{0, 0}, {64, 1}, {128, 2}, {160, 0}});
        size_t c3, other, second;
        EXPECT(arena_take_straddling(a, round_up(10 * GiB), &c3));
        EXPECT(class_at(a, c3) != class_at(a, c3 + 10 * GiB - 1));                       // it crosses ...
        EXPECT(class_at(a, c3 + 5 * GiB - 3 * MiB) != class_at(a, c3 + 5 * GiB + 3 * MiB));  // ... at its middle
        EXPECT(kmers_arena_straddles(a, a.base + c3, 10 * GiB));
        EXPECT(!kmers_arena_straddles(a, a.base + c3, 4 * GiB));                         // only its first 4 GiB written: one class
        EXPECT(!kmers_arena_straddles(a, a.base + c3 + 300 * GiB, GiB) && !kmers_arena_straddles(a, nullptr, GiB));
        EXPECT(take(a, 8 * GiB, &other));
        EXPECT(!kmers_arena_straddles(a, a.base + other, 8 * GiB));                      // an ordinary block lies inside one run
        EXPECT(arena_take_straddling(a, round_up(10 * GiB), &second) && second != c3);   // the next boundary
        EXPECT(kmers_arena_straddles(a, a.base + second, 10 * GiB));
        // 90 GiB: no boundary has 45 GiB of one run on either side, but [19, 109) across the first one is free and is taken
        // (its halves reach into a third run: the penalised candidate, the only one)
        size_t big;
        give(a, c3); give(a, other); give(a, second);
        EXPECT(arena_take_straddling(a, round_up(90 * GiB), &big) && big + 45 * GiB <= 64 * GiB + 2 * MiB && big + 45 * GiB + 2 * MiB >= 64 * GiB);
        give(a, big);
        // nothing free around any boundary: refused (the caller falls back to arena_take)
        size_t f0, f1, f2, none;
        EXPECT(take(a, 60 * GiB, &f0) && take(a, 60 * GiB, &f1) && take(a, 60 * GiB, &f2));
        EXPECT(!arena_take_straddling(a, round_up(30 * GiB), &none));
        kmers_arena flat = fresh(64 * GiB);                                              // no map / one run: refused
        EXPECT(!arena_take_straddling(flat, round_up(GiB), &none));
        set_map(flat, {{0, 0}});
        EXPECT(!arena_take_straddling(flat, round_up(GiB), &none) && !kmers_arena_straddles(flat, flat.base, GiB));
    }
    {  // ---- two blocks longer than any run on a FINE-GRAINED map (a box of round 3: runs of 4-32 GiB): the second block goes where
       //      its runs pair best with the first block's, inside the free range if that beats both of its ends
        kmers_arena a = fresh(212 * GiB);
        set_map(a, {{0, 0}, {32, 1}, {48, 2}, {64, 0}, {96, 2}, {104, 1}, {120, 2}, {136, 0}, {152, 2}, {160, 0}, {164, 1}, {188, 2}, {200, 0}, {204, 1}});
        const size_t need = round_up((size_t)80 * 1000 * 1000 * 1000);
        size_t first, second;
        EXPECT(take(a, need, &first) && first == 0);
        EXPECT(take(a, need, &second) && second >= need);
        auto differ = [&](size_t off) {  // fraction of the block whose class differs from the first block's at the same relative place
            int d = 0;
            for (int i = 0; i < 64; ++i) {
                const size_t t = (size_t)((2 * i + 1) * (double)need / 128.0);
                d += class_at(a, first + t) != class_at(a, off + t);
            }
            return d / 64.0;
        };
        const size_t top = (212 * GiB - need) / GRANULE * GRANULE;
        EXPECT(differ(second) >= differ(need) && differ(second) >= differ(top));
        EXPECT(differ(second) >= 0.8);
        std::printf("fine-grained map: second block at %.1f GiB, classes differ over %.0f %% of it (bottom end %.0f %%, top end %.0f %%)\n",
                    second / (double)GiB, 100 * differ(second), 100 * differ(need), 100 * differ(top));
        size_t src;
        EXPECT(take(a, 5 * GiB, &src));   // the sequence still fits
    }
    {  // ---- the same on a COARSE map where one run could hold the first block: A 64, B 96, C 52 GiB (a box of round 3)
        kmers_arena a = fresh(212 * GiB);
        set_map(a, {{0, 0}, {64, 1}, {160, 2}});
        const size_t need = round_up((size_t)80 * 1000 * 1000 * 1000);
        size_t first, second, src;
        EXPECT(take(a, need, &first) && first == 0);          // not inside B: the bottom of the block
        EXPECT(take(a, need, &second));
        int same = 0;
        for (int i = 0; i < 64; ++i) {
            const size_t t = (size_t)((2 * i + 1) * (double)need / 128.0);
            same += class_at(a, first + t) == class_at(a, second + t);
        }
        EXPECT(same == 0);                                     // [0, 74.5) = A, B against [138.5, 213) = B, C: never the same class
        EXPECT(take(a, 5 * GiB, &src));
    }
    {  // ---- invariants under a random sequence of requests (with a fragmented map)
        kmers_arena a = fresh(96 * GiB + 6 * MiB);
        set_map(a, {{0, 0}, {16, 1}, {20, 0}, {40, 2}, {72, 1}, {80, 0}});
        std::mt19937_64 rng(12345);
        std::vector<size_t> live;
        for (int step = 0; step < 20000; ++step) {
            if (live.empty() || rng() % 3) {
                const size_t sizes[] = {1, 3 * MiB, 100 * MiB, GiB, 5 * GiB, 17 * GiB, 40 * GiB};
                size_t off;
                if (take(a, sizes[rng() % 7], &off)) live.push_back(off);
            } else {
                const size_t i = rng() % live.size();
                give(a, live[i]);
                live.erase(live.begin() + (long)i);
            }
            if (step % 500 == 0) {
                size_t covered = 0, prev_end = 0;
                std::map<size_t, size_t> all = a.free_ranges;
                for (auto &u : a.used) {
                    EXPECT(all.find(u.first) == all.end());
                    all[u.first] = u.second;
                }
                for (auto &r : all) {
                    EXPECT(r.first == prev_end && r.first % GRANULE == 0 && r.second % GRANULE == 0 && r.second > 0);  // no gap, no overlap
                    prev_end = r.first + r.second;
                    covered += r.second;
                }
                EXPECT(covered == a.bytes);
                for (auto it = a.free_ranges.begin(); it != a.free_ranges.end(); ++it) {  // free neighbours are always merged
                    auto nx = std::next(it);
                    if (nx != a.free_ranges.end()) EXPECT(it->first + it->second < nx->first);
                }
            }
        }
        for (size_t off : live) give(a, off);
        EXPECT(a.used.empty() && a.free_ranges.size() == 1 && a.free_ranges.begin()->second == a.bytes);
    }
    std::printf(failures ? "FAILURES %d\n" : "arena placement ok (%d)\n", failures);
    return failures != 0;
}
