// stripe_pool_check.cpp -- the pure logic of the striped pool (kmers.jl_amd/csrc/stripe_pool.hpp) on the CPU:
//   g++ -std=c++17 -O1 -I kmers.jl_amd/csrc -o /tmp/stripe_pool_check tests/c/stripe_pool_check.cpp && /tmp/stripe_pool_check
#include <cstdio>
#include <cstdlib>
#include <random>

#include "stripe_pool.hpp"

using namespace kmers::pool;

#define REQUIRE(x)                                                          \
    do {                                                                    \
        if (!(x)) {                                                         \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #x);    \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static State make(const std::vector<int> &unit_classes) {  // one class per HALF unit
    State s;
    for (size_t h = 0; h < unit_classes.size(); ++h) {
        if (h % 2 == 0) {
            Unit u;
            u.first_chunk = (uint32_t)s.chunks.size();
            s.units.push_back(u);
        }
        for (uint32_t i = 0; i < UNIT_CHUNKS / 2; ++i) {
            Chunk c;
            c.unit = (uint32_t)s.units.size() - 1;
            c.cls = (uint8_t)unit_classes[h];
            c.handle = &s;
            s.free_list[c.cls].push_back((uint32_t)s.chunks.size());
            s.chunks.push_back(c);
        }
    }
    s.held_bytes = s.chunks.size() * CHUNK_BYTES;
    return s;
}

int main() {
    // water-filling
    {
        size_t fr[N_LISTS] = {100, 10, 3, 0, 0}, c[N_LISTS];
        REQUIRE(pick_counts(fr, 20, c));
        REQUIRE(c[0] + c[1] + c[2] == 20 && c[2] == 3 && c[1] >= 8 && c[0] <= 9);
        REQUIRE(balanced(fr, 20));
        REQUIRE(pick_counts(fr, 60, c) && c[0] == 47 && c[1] == 10 && c[2] == 3);
        REQUIRE(!balanced(fr, 60));  // 47 of 60 from one class: neighbours of one class cannot be avoided
        REQUIRE(!pick_counts(fr, 114, c));
        REQUIRE(pick_counts(fr, 113, c) && c[0] == 100);
        size_t one[N_LISTS] = {8, 0, 0, 0, 0};
        REQUIRE(pick_counts(one, 1, c) && balanced(one, 1) && !balanced(one, 2));
    }
    // the order: neighbours differ whenever no class has more than half; ties go round-robin
    {
        size_t c[N_LISTS] = {4, 4, 4, 0, 0};
        auto o = stripe_order(c, 12);
        REQUIRE(o.size() == 12);
        for (size_t i = 0; i < 12; ++i) REQUIRE(o[i] == i % 3);
        size_t d[N_LISTS] = {5, 3, 1, 0, 0};
        o = stripe_order(d, 9);
        for (size_t i = 1; i < o.size(); ++i) REQUIRE(o[i] != o[i - 1]);
        size_t e[N_LISTS] = {7, 2, 0, 0, 0};  // cannot alternate: still every chunk is placed
        o = stripe_order(e, 9);
        REQUIRE(o.size() == 9);
    }
    // random maps: take / give keep the books, blocks alternate when the pool allows it
    std::mt19937 rng(5);
    for (int trial = 0; trial < 200; ++trial) {
        std::vector<int> halves(2 * (2 + rng() % 30));
        int cls = 0;
        for (auto &h : halves) {
            if (rng() % 4 == 0) cls = (int)(rng() % 3);
            h = cls;
        }
        State s = make(halves);
        const size_t total = s.chunks.size();
        std::vector<std::vector<uint32_t>> out;
        size_t used = 0;
        for (int k = 0; k < 6; ++k) {
            const size_t n = 1 + rng() % (total / 3 + 1);
            size_t fr[N_LISTS];
            for (int i = 0; i < N_LISTS; ++i) fr[i] = s.free_list[i].size();
            const bool bal = balanced(fr, n);
            auto ids = take(s, n);
            if (used + n > total) {
                REQUIRE(ids.empty());
                continue;
            }
            REQUIRE(ids.size() == n);
            used += n;
            REQUIRE(s.in_use_bytes == used * CHUNK_BYTES);
            for (auto id : ids) REQUIRE(s.chunks[id].in_use);
            if (bal && n >= 2) REQUIRE(alternation_of(s, ids) == 1.f);
            out.push_back(ids);
        }
        for (auto &ids : out) give(s, ids);
        REQUIRE(s.in_use_bytes == 0);
        size_t fr = 0;
        for (auto &l : s.free_list) fr += l.size();
        REQUIRE(fr == total);
        for (auto &u : s.units) REQUIRE(u.in_use == 0);
    }
    // block lookup
    {
        State s = make({0, 1});
        char *base = reinterpret_cast<char *>((size_t)1 << 40);
        Block b;
        b.bytes = 4 * CHUNK_BYTES;
        s.blocks[base] = b;
        REQUIRE(block_of(s, base, 1) && block_of(s, base + 4 * CHUNK_BYTES - 1, 1) && !block_of(s, base + 4 * CHUNK_BYTES, 1));
        REQUIRE(!block_of(s, base - 1, 1) && !block_of(s, base + CHUNK_BYTES, 4 * CHUNK_BYTES));
    }
    std::puts("stripe_pool_check: ok");
    return 0;
}
