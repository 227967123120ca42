// class_pool_check.cpp -- the pure logic of the device's class pool (kmers.jl_amd/csrc/class_pool.hpp) on the CPU:
//   g++ -std=c++17 -O1 -I kmers.jl_amd/csrc -o /tmp/class_pool_check tests/c/class_pool_check.cpp && /tmp/class_pool_check
#include <cstdio>
#include <cstdlib>
#include <random>

#include "class_pool.hpp"

using namespace kmers::pool;

#define REQUIRE(x)                                                          \
    do {                                                                    \
        if (!(x)) {                                                         \
            std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, #x);    \
            std::exit(1);                                                   \
        }                                                                   \
    } while (0)

static std::vector<uint8_t> runs(std::initializer_list<std::pair<int, int>> r) {
    std::vector<uint8_t> v;
    for (auto &p : r) v.insert(v.end(), (size_t)p.second, (uint8_t)p.first);
    return v;
}
static size_t n_runs(const std::vector<uint8_t> &v) {
    size_t n = v.empty() ? 0 : 1;
    for (size_t i = 1; i < v.size(); ++i) n += v[i] != v[i - 1];
    return n;
}

static Block block_of_classes(const std::vector<uint8_t> &classes, size_t req_bytes = 0) {
    Block b;
    b.classes = classes;
    b.chunks.resize(classes.size());
    b.bytes = classes.size() * CHUNK_BYTES;
    b.req_bytes = req_bytes ? req_bytes : b.bytes;
    return b;
}
static const size_t G = CHUNK_BYTES;

int main() {
    float q = 0.f;
    // no partner: as pure as the stock allows, the fullest class first
    {
        size_t fr[N_LISTS] = {5, 9, 2, 0};
        auto s = plan(fr, 8 * G, nullptr, ROLE_DEFAULT, &q);
        REQUIRE(s == runs({{1, 8}}) && q == 1.f);
        s = plan(fr, 12 * G - 5, nullptr, ROLE_DEFAULT, &q);
        REQUIRE(s.size() == 12 && n_runs(s) == 2 && s[0] == 1);
        REQUIRE(plan(fr, 17 * G, nullptr, ROLE_DEFAULT, &q).empty());
        REQUIRE(plan(fr, 16 * G, nullptr, ROLE_DEFAULT, &q).size() == 16);
    }
    // the second array of a launch: another class at every position
    {
        size_t fr[N_LISTS] = {20, 8, 3, 0};
        Block a = block_of_classes(runs({{0, 8}}), 8 * G - 400000000);
        auto b = plan(fr, 8 * G - 400000000, &a, ROLE_DEFAULT, &q);
        REQUIRE(b == runs({{1, 8}}) && q == 1.f);
        // the partner spans three classes; the stock is exactly what a perfect answer needs, and the greedy choice alone gets stuck
        size_t eq[N_LISTS] = {25, 25, 25, 0};
        Block p3 = block_of_classes(runs({{0, 25}, {1, 25}, {2, 25}}));
        auto s = plan(eq, 75 * G, &p3, ROLE_DEFAULT, &q);
        REQUIRE(s.size() == 75 && q == 1.f && n_runs(s) == 3);
        // arrays of different lengths (two-word kmers beside one-word hashes): positions are relative
        auto half = plan(eq, 75 * G / 2, &p3, ROLE_DEFAULT, &q);
        REQUIRE(half.size() == 38 && q >= 0.9f);
        // not enough of the other classes: as many positions as possible, the rest from what there is
        size_t poor[N_LISTS] = {30, 3, 0, 0};
        auto r = plan(poor, 8 * G, &a, ROLE_DEFAULT, &q);
        REQUIRE(r.size() == 8 && q == 3.f / 8.f);
        // unknown chunks are better than the forbidden class, but count for nothing
        size_t unk[N_LISTS] = {30, 0, 0, 8};
        r = plan(unk, 8 * G, &a, ROLE_DEFAULT, &q);
        REQUIRE(r == runs({{CLASS_UNKNOWN, 8}}) && q == 0.f);
    }
    // a second block to stay away from if the stock allows: the third class for the sequence; given up before the first
    {
        size_t fr[N_LISTS] = {10, 10, 10, 0};
        Block a = block_of_classes(runs({{0, 8}})), b = block_of_classes(runs({{1, 8}}));
        auto t = plan(fr, 2 * G, &b, ROLE_DEFAULT, &q, &a);
        REQUIRE(t == runs({{2, 2}}) && q == 1.f);
        size_t no_c[N_LISTS] = {10, 10, 0, 0};
        t = plan(no_c, 2 * G, &b, ROLE_DEFAULT, &q, &a);
        REQUIRE(t == runs({{0, 2}}) && q == 1.f);        // (quality counts the partner only)
        // kmers, then a small block, then the hashes: the hashes are planned beside the KMERS
        State s;
        char *base = reinterpret_cast<char *>((size_t)1 << 40);
        Block small = block_of_classes(runs({{1, 1}}));
        a.serial = 1;
        small.serial = 2;
        s.blocks[base] = a;
        s.blocks[base + 16 * G] = small;
        const Block *other = nullptr;
        const Block *p = partner_block(s, 8 * G, &other);
        REQUIRE(p->serial == 1 && other->serial == 2);
        auto h = plan(fr, 8 * G, p, ROLE_DEFAULT, &q, other);
        REQUIRE(h == runs({{2, 8}}) && q == 1.f);
    }
    // the only output of a launch: second half against first half, in BYTES of the array (not in chunks of the block)
    {
        size_t fr[N_LISTS] = {20, 8, 3, 0};
        auto s = plan(fr, 10 * G, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s == runs({{0, 5}, {1, 5}}) && q == 1.f);
        // 9.31 GiB (1.25 G kmers): the middle of the array is inside chunk 4, whose middle lies in the first half
        s = plan(fr, (size_t)10000000000, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s == runs({{0, 5}, {1, 5}}) && q == 1.f);
        // 2.48 GiB (C5: 333 M kmers): one chunk in front, two behind
        s = plan(fr, (size_t)2666666664, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s == runs({{0, 1}, {1, 2}}) && q == 1.f);
        size_t two[N_LISTS] = {6, 5, 0, 0};
        s = plan(two, 10 * G, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s.size() == 10 && q == 1.f);
        size_t one[N_LISTS] = {12, 0, 0, 0};
        s = plan(one, 10 * G, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s.size() == 10 && q == 0.f);
        size_t big[N_LISTS] = {40, 30, 20, 0};  // no class covers a half AND leaves the rest to the others... two do
        s = plan(big, 80 * G, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s.size() == 80 && q == 1.f);
        size_t tight[N_LISTS] = {30, 30, 20, 0};  // no class holds a half of 40: the first half is mixed, the second still differs everywhere
        s = plan(tight, 80 * G, nullptr, ROLE_LONE_OUTPUT, &q);
        REQUIRE(s.size() == 80 && q == 1.f);
    }
    // random stock and partners: never more chunks of a class than there are; perfect whenever the other classes hold enough
    std::mt19937 rng(7);
    for (int trial = 0; trial < 3000; ++trial) {
        size_t fr[N_LISTS];
        size_t total = 0;
        for (int c = 0; c < N_LISTS; ++c) total += fr[c] = c < 3 ? rng() % 40 : rng() % 3;
        if (!total) continue;
        const size_t n = 1 + rng() % total;
        const size_t bytes = n * G - (rng() % 2 ? rng() % (G / 2) : 0);
        Block partner;
        if (rng() % 4) {
            const size_t m = 1 + rng() % 60;
            std::vector<uint8_t> cl;
            uint8_t c = (uint8_t)(rng() % 3);
            for (size_t i = 0; i < m; ++i) {
                if (rng() % 6 == 0) c = (uint8_t)(rng() % 3);
                cl.push_back(c);
            }
            partner = block_of_classes(cl, m * G - rng() % (G / 2));
        }
        const int role = rng() % 3 == 0 ? ROLE_LONE_OUTPUT : ROLE_DEFAULT;
        auto s = plan(fr, bytes, partner.classes.empty() ? nullptr : &partner, role, &q);
        REQUIRE(s.size() == n);
        size_t used[N_LISTS] = {};
        for (uint8_t c : s) {
            REQUIRE(c < N_LISTS);
            ++used[c];
        }
        for (int c = 0; c < N_LISTS; ++c) REQUIRE(used[c] <= fr[c]);
        if (role == ROLE_DEFAULT && !partner.classes.empty()) {
            bool plenty = true;  // every forbidden class can be avoided everywhere
            for (int f = 0; f < 3; ++f) plenty &= total - fr[f] - fr[CLASS_UNKNOWN] >= n;
            if (plenty) REQUIRE(q == 1.f);
        }
    }
    // take / give keep the books
    {
        State s;
        for (uint32_t i = 0; i < 12; ++i) {
            Chunk c;
            c.cls = (uint8_t)(i % 3);
            c.handle = &s;
            s.chunks.push_back(c);
            s.free_list[c.cls].push_back(i);
        }
        size_t fr[N_LISTS];
        for (int c = 0; c < N_LISTS; ++c) fr[c] = s.free_list[c].size();
        auto seq = plan(fr, 6 * G, nullptr, ROLE_LONE_OUTPUT, nullptr);
        auto ids = take(s, seq);
        REQUIRE(ids.size() == 6 && s.in_use_bytes == 6 * CHUNK_BYTES);
        for (size_t i = 0; i < 6; ++i) REQUIRE(s.chunks[ids[i]].cls == seq[i] && s.chunks[ids[i]].in_use);
        give(s, ids);
        REQUIRE(s.in_use_bytes == 0);
        size_t total = 0;
        for (auto &l : s.free_list) total += l.size();
        REQUIRE(total == 12);
    }
    // arrays inside blocks
    {
        State s;
        char *base = reinterpret_cast<char *>((size_t)1 << 40);
        Block a, b, c;
        a.bytes = b.bytes = 8 * CHUNK_BYTES;
        a.classes = runs({{0, 8}});
        b.classes = runs({{1, 8}});
        c.bytes = 8 * CHUNK_BYTES;
        c.classes = runs({{0, 4}, {2, 4}});
        a.chunks.resize(8);
        b.chunks.resize(8);
        c.chunks.resize(8);
        a.serial = 1;
        b.serial = 2;
        c.serial = 3;
        s.blocks[base] = a;
        s.blocks[base + 16 * CHUNK_BYTES] = b;
        s.blocks[base + 32 * CHUNK_BYTES] = c;
        const size_t bytes = 8 * CHUNK_BYTES - 12345;
        REQUIRE(arrays_differ(s, base, bytes, base + 16 * CHUNK_BYTES, bytes) == 1.f);
        REQUIRE(arrays_differ(s, base, bytes, base + 4096, bytes - 4096) == 0.f);
        REQUIRE(arrays_differ(s, base, bytes, base + 8 * CHUNK_BYTES, 16) == -1.f);  // not inside a block
        REQUIRE(halves_differ(s, base, bytes) == 0.f && halves_differ(s, base + 32 * CHUNK_BYTES, bytes) == 1.f);
        const Block *other = nullptr;
        REQUIRE(partner_block(s, 8 * G, &other)->serial == 3 && other && other->serial == 2);
        REQUIRE(partner_block(s, 100 * G, &other)->serial == 3);   // nothing comparable: the most recent
        REQUIRE(block_of(s, base, 1) && !block_of(s, base + 8 * CHUNK_BYTES, 1) && !block_of(s, base - 1, 1));
    }
    // the cache of freed blocks and the bound on the hoard (round 6): a collect loop re-uses its blocks, a request beside a new
    // partner takes the cached block that suits it, nothing else does
    {
        State s;
        char *base = reinterpret_cast<char *>((size_t)1 << 40);
        auto make = [&](std::vector<uint8_t> classes, size_t req, int role, float quality) {
            Block b = block_of_classes(classes, req);
            b.role = role;
            b.quality = quality;
            b.serial = ++s.serial;
            return b;
        };
        const size_t bytes = 8 * G - 12345;
        size_t off = 0, pb = 0;
        REQUIRE(block_chunks(bytes, ROLE_DEFAULT, &off, &pb) == 8 && off == 0 && pb == bytes);
        REQUIRE(block_chunks(10 * G - 7 * (G / 10), ROLE_LONE_OUTPUT, &off, &pb) == 10 && pb == 10 * G && off > 0 && off < G);
        REQUIRE(block_chunks(G / 2, ROLE_LONE_OUTPUT, &off, &pb) == 1 && off == 0);
        // two arrays out, both freed: k (class 0) then h (class 1)
        s.blocks[base] = make(runs({{0, 8}}), bytes, ROLE_DEFAULT, 1.f);
        s.blocks[base + 16 * G] = make(runs({{1, 8}}), bytes, ROLE_DEFAULT, 1.f);
        s.in_use_bytes = 16 * G;
        for (char *p : {base, base + 16 * G}) {
            ++s.tick;
            auto it = s.blocks.find(p);
            Block b = std::move(it->second);
            s.blocks.erase(it);
            to_cache(s, p, std::move(b));
        }
        REQUIRE(s.in_use_bytes == 0 && s.cached_bytes == 16 * G && s.cached.size() == 2);
        // the next k: no partner, the most recently freed block of its shape
        auto hit = find_cached(s, bytes, ROLE_DEFAULT, nullptr);
        REQUIRE(hit != s.cached.end() && hit->first == base + 16 * G);
        REQUIRE(find_cached(s, 4 * G, ROLE_DEFAULT, nullptr) == s.cached.end());        // another number of chunks
        REQUIRE(find_cached(s, bytes, ROLE_LONE_OUTPUT, nullptr) == s.cached.end());    // another role
        const char *hit_base = hit->first;  // (the iterator dies with the cache entry)
        s.blocks[hit_base] = from_cache(s, hit);
        REQUIRE(s.in_use_bytes == 8 * G && s.cached_bytes == 8 * G);
        // the next h beside it: the cached block of the OTHER class
        const Block *partner = partner_block(s, bytes, nullptr);
        REQUIRE(partner && partner->classes[0] == 1);
        hit = find_cached(s, bytes, ROLE_DEFAULT, partner);
        REQUIRE(hit != s.cached.end() && hit->first == base && hit->second.classes[0] == 0);
        // a cached block in the partner's own class does not qualify ... unless the plan it was assembled by was no better
        s.cached[base + 32 * G] = make(runs({{1, 8}}), bytes, ROLE_DEFAULT, 1.f);
        s.cached_bytes += 8 * G;
        s.cached.erase(base);
        s.cached_bytes -= 8 * G;
        REQUIRE(find_cached(s, bytes, ROLE_DEFAULT, partner) == s.cached.end());
        s.cached[base + 32 * G].quality = 0.f;
        REQUIRE(find_cached(s, bytes, ROLE_DEFAULT, partner) != s.cached.end());
        // the hoard: free chunks + yardsticks beyond max(4 chunks, a quarter of the blocks) go back
        State h;
        h.n_classes = 3;
        for (uint32_t i = 0; i < 40; ++i) {
            Chunk c;
            c.cls = (uint8_t)(i % 3);
            h.chunks.push_back(c);
            h.free_list[c.cls].push_back(i);
        }
        REQUIRE(hoard_excess(h) == 43 - 4);
        h.in_use_bytes = 16 * G;
        h.cached_bytes = 16 * G;
        REQUIRE(hoard_excess(h) == 43 - 8);
        h.in_use_bytes = 400 * G;
        REQUIRE(hoard_excess(h) == 0);
    }
    std::puts("class_pool_check: ok");
    return 0;
}
