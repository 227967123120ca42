// unambiguous_kernel.hpp -- UnambiguousKmers (src/iterators/UnambiguousKmers.jl:59-148) in ONE pass over the source:
// every window of K unambiguous symbols together with its 1-based start index, in the reference's order.
//
// The reference walks the sequence with a `remaining` counter that every ambiguous symbol resets (:140-146).  Here a
// window is kept iff the K bits of a one-bit-per-symbol ambiguity stream (staged in LDS next to the 2-bit code stream)
// are all zero, which selects exactly the same windows.  The output position of a kept window is the number of kept
// windows before it -- a prefix sum over the whole sequence -- and it is resolved INSIDE the emitting kernel:
//
//   tile     a workgroup draws a ticket (tiles are numbered in the order workgroups start), stages its source words once,
//            resolves all of its candidate starts bit-parallel (64 starts per lane: the "good" bits AND-ed with themselves
//            shifted by 1, 2, 4, ...), and publishes its kept count as an AGGREGATE in its tile descriptor;
//   look-back   one wavefront then sums the descriptors of the preceding tiles, nearest first, 256 per step, until it meets
//            one that already holds an inclusive PREFIX (decoupled look-back: it never waits for a predecessor's prefix,
//            only for aggregates, which every started tile publishes unconditionally -- no circular wait), publishes its
//            own inclusive prefix and hands the exclusive one to the workgroup;
//   emit     every wavefront lists the kept starts of 1024-2048 candidates at a time in LDS (their order is the reference's)
//            and works the list off with every lane busy: window cut + contiguous stores.  A stretch with nothing dropped
//            (real sequence outside its N blocks) skips the list: two kmers per lane, 16-byte stores.
//   pipeline the grid is persistent (four workgroups per CU) and a workgroup runs the first step of its NEXT tile before the
//            last two of the current one, so that an aggregate is out microseconds after its ticket.  Between the two a
//            tile's keep mask and prefix counts stay in the registers of the lanes that resolved them; its codes in LDS.
//
// A descriptor is ONE 64-bit word (status in the top two bits, count below) moved with relaxed agent-scope atomics, so
// no fence is needed: the word is the whole hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, "8-B agent atomics
// both sides").  The source is read once; there is no count pass, no scan launch and no host round trip before the emit.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

// 49152-start tiles, four workgroups per CU (36 KiB of LDS each): late in round 3, with the outputs in two region classes, fewer
// and longer tiles win -- K = 31 0.71 -> 0.73, the stride-3 lattice 0.58 -> 0.61 of 8 TB/s; 6 x 32768 (rounds 2-3), 5 x 40960 and
// 3 x 65536 run the same, 7 x 28672, 8 x 24576 and 2 x 98304 lose (profiles/r03_tuning.md, tools/r3_unamb_occupancy.sh)
#ifndef KMERS_UTILE_MAX
#define KMERS_UTILE_MAX 49152
#endif
constexpr uint32_t UTILE_MAX = KMERS_UTILE_MAX;  // candidate starts per tile (a multiple of 1024), at most
#ifndef KMERS_UROUND
#define KMERS_UROUND 1024
#endif
constexpr uint32_t UROUND = KMERS_UROUND;  // starts per wavefront round (16 or 8 per lane)
constexpr uint32_t USLICE = UROUND / 64;  // consecutive starts per lane
// A round whose kept starts fit the wavefront's list takes 2048 candidate starts at once (32 per lane): with 14-28 % of the
// starts kept, a 1024-start round lists 140-290 elements and fills its last 128-element store pass badly.  (4096-start rounds
// with 2048-entry lists ran the same and cost 8 KiB of LDS more; the kernel's time falls with every resident workgroup,
// profiles/r02_tuning.md section 6)
#ifndef KMERS_ULONG
#define KMERS_ULONG 2048
#endif
#ifndef KMERS_ULIST
#define KMERS_ULIST 1024
#endif
constexpr uint32_t ULONG = KMERS_ULONG, ULIST = KMERS_ULIST;
// The emitting path stores in FRAMES: 128 consecutive output elements, aligned to 128 in the OUTPUT index (lane l of a frame owns
// elements 2l and 2l + 1: one 16-byte store per lane and array, 1 KiB = eight whole 128-byte lines per wave store).  What a
// round lists beyond its last whole frame stays at the head of the list for the next round (fewer than UFRAME entries), so a
// wavefront issues partial store instructions only at the two ends of its quarter of the tile, not at the ends of every round
// (with 14-28 % of the starts kept a 2048-start round used to end in a store pass that was 25-50 % full, plus up to four
// one-lane stores for odd heads and tails: 45-65 % of the bytes a store instruction can carry, profiles/r03_unamb.md).
constexpr uint32_t UFRAME = 128;
constexpr uint32_t ULSTRIDE = ULIST + UFRAME;  // list entries per wavefront: a round's kept starts behind the carried ones
static_assert(ULIST >= UROUND, "a short round may keep every one of its starts");
static_assert(ULONG % UROUND == 0 && (ULONG == 1024 || ULONG == 2048 || ULONG == 4096) && 64 % USLICE == 0, "a lane's slice of a round lies in one keep-mask qword");
#ifndef KMERS_UNAMB_WGS
#define KMERS_UNAMB_WGS 4
#endif
constexpr int UNAMB_EMIT_WGS = KMERS_UNAMB_WGS;  // workgroups per CU of the emitting mode (36 KiB of LDS each)
constexpr uint64_t DESC_VALUE = (1ull << 62) - 1ull;
constexpr uint64_t DESC_AGGREGATE = 1ull << 62, DESC_PREFIX = 2ull << 62;
constexpr int LOOKBACK = 4;               // descriptors per lane and look-back step (256 tiles per step)
// Every spin is bounded (MI355X_MICROARCH.md, correctness boundaries): a look-back polls a missing aggregate at most
// SPIN_LIMIT times (well over a second; a tile publishes its aggregate microseconds after its ticket), then raises the abort
// flag, which every other look-back checks every SPIN_CHECK polls: the kernel drains and the host reports KMERS_E_HIP
// instead of hanging the device.
#ifdef KMERS_TEST_ABORT
// test build (kmers_jl_amd/build.py: build_test_abort): tile 1 never publishes its aggregate, so the look-back of tile 2 runs
// into the limit, raises the abort flag, every workgroup drains and the host reports KMERS_E_HIP -- the path that cannot be
// provoked on a healthy device, exercised once (tests/test_gpu_parity.py::test_unambiguous_lookback_gives_up_instead_of_hanging)
constexpr uint32_t SPIN_CHECK = 64, SPIN_LIMIT = 1u << 12;
#else
constexpr uint32_t SPIN_CHECK = 1024, SPIN_LIMIT = 1u << 22;
#endif

enum UMode { UMODE_EMIT = 0, UMODE_COUNT = 1, UMODE_XOR = 2 };

struct UnambArgs {
    const uint64_t *src;
    uint64_t first_bit;
    uint64_t n_cand;               // candidate starts = n_bases - K + 1
    uint64_t n_bases;              // SRC_BITS == 8: every byte below n_bases is inspected (UnambiguousKmers.jl:117-123)
    uint64_t n_tiles;
    unsigned long long *desc;      // EMIT: [n_tiles] tile descriptors, zeroed before the launch
    unsigned long long *ticket;    // EMIT: zeroed; tile ids are drawn in the order workgroups start
    unsigned long long *abort_flag;  // EMIT: zeroed; set if a look-back ever waits longer than it possibly can (see SPIN_LIMIT)
    unsigned long long *total;     // COUNT: += kept starts;  XOR: ^= head words of the kept kmers
    uint64_t *out_kmers;           // nullable
    long long *out_starts;         // nullable
    uint64_t capacity;             // elements the outputs hold: nothing is stored at or beyond it
    uint64_t index_origin;
    unsigned long long *err_slot;  // SRC_BITS == 8: first invalid byte (0xff in the table) -> EncodeError
    uint32_t k;
    uint32_t stride;               // keep windows with (start0 % stride) == 0
    uint32_t tile_starts;          // multiple of 1024, <= UTILE_MAX
    uint32_t ascii_table;          // SRC_BITS == 8: ASCII_TABLE_SKIPPING = the reference's ASCII_SKIPPING_LUT (common.jl:22-32)
    uint32_t tuples;               // 1: out_kmers receives Tuple{Kmer,Int} elements (N + 1 words each), out_starts unused
    uint32_t vec16;                // out_kmers / out_starts are 16-byte aligned (16-byte stores allowed)
    uint32_t n_words;              // words per kmer (the N = 0 instantiation takes it at run time: kmers of more than four words)
    uint64_t *stamps;              // diagnostic builds (-DKMERS_STAMPS) only: 8 s_memrealtime stamps per tile
};

__device__ __forceinline__ unsigned long long desc_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void desc_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// bit j of the result: K unambiguous symbols begin at symbol (bit + j) of the flag stream (one bit per symbol, set =
// ambiguous).  The window of start j spans flag bits [j, j + K): up to 64 + 127 bits for 64 starts and K <= 128.
__device__ __forceinline__ uint64_t keep_qword(const uint64_t *amb, uint32_t bit, uint32_t k) {
    const uint32_t Q = bit >> 6, sft = bit & 63u;
    uint64_t lo = ~funnel64(amb[Q], amb[Q + 1], sft), mid = ~funnel64(amb[Q + 1], amb[Q + 2], sft);
    if (k <= 65) {
        uint32_t have = 1;
        while (have < k) {
            const uint32_t step = have < k - have ? have : k - have;  // 1..32: good[j..j+have) & good[j+step..j+step+have)
            lo &= (lo >> step) | ((mid << 1) << (63u - step));
            mid &= mid >> step;  // (the bits shifted in from beyond `mid` are never needed: have + step <= 65)
            have += step;
        }
        return lo;
    }
    if (k > 129) {
        // kmers of more than four words (an edge path): start j is kept iff the flag bits [j, j + K) are all zero, tested word
        // by word; every lane walks the 64 + K bits behind its 64 starts once, keeping the distance to the next flag
        uint64_t keep = 0;
        uint32_t clear = 0;  // unflagged symbols directly behind the position being visited (walking downwards)
        for (int p = (int)(63u + k) - 1; p >= 0; --p) {  // symbol bit + p, from the far end of the last window down to start 0
            const uint32_t b = bit + (uint32_t)p;
            clear = ((amb[b >> 6] >> (b & 63u)) & 1ull) ? 0u : clear + 1u;
            if (p < 64 && clear >= k) keep |= 1ull << p;
        }
        return keep;
    }
    uint64_t hi = ~funnel64(amb[Q + 2], amb[Q + 3], sft);
    uint32_t have = 1;
    while (have < k) {
        uint32_t step = have < k - have ? have : k - have;
        if (step > 32u) step = 32u;
        lo &= (lo >> step) | ((mid << 1) << (63u - step));
        mid &= (mid >> step) | ((hi << 1) << (63u - step));
        hi &= hi >> step;
        have += step;
    }
    return lo;
}

// This kernel only ever needs FORWARD kmers, so it stages the 2-bit codes of a tile in KMER order (cut_fw, stream_kernel.hpp).

template <int SRC_BITS, int N, int UMODE>
__global__ __launch_bounds__(BLOCK, UMODE == UMODE_EMIT ? UNAMB_EMIT_WGS : 4) void unambiguous_kernel(const UnambArgs a) {
    constexpr bool EMIT = UMODE == UMODE_EMIT;
    constexpr uint32_t NBUF = EMIT ? 2u : 1u;  // EMIT resolves tile n+1 before it emits tile n: two sets of tile state
    constexpr uint32_t STREAM_QWORDS = (UTILE_MAX + 128 + 64) / 32 + 4;   // 2-bit codes of the tile + its K-1 overlap
    constexpr uint32_t AMB_QWORDS = (UTILE_MAX + 128 + 64) / 64 + 6;
    constexpr uint32_t MAXQ = UTILE_MAX / 64;
    __shared__ uint64_t lds2[NBUF][STREAM_QWORDS];
    // The keep mask of a tile and the kept starts before each of its 64-start qwords never go to LDS: thread t resolves
    // qwords QPT*t .. QPT*t + QPT-1 and keeps them in registers until the tile is emitted (one whole front later); wavefront
    // w then works on exactly the qwords its own lanes hold, and fetches what a lane needs with a lane shuffle.
    constexpr uint32_t QPT = (MAXQ + (uint32_t)BLOCK - 1u) / (uint32_t)BLOCK;  // qwords per thread (two with 32768-start tiles)
    constexpr uint32_t WQ = 64u * QPT;                                          // qwords per wavefront
    struct TileRegs {
        uint64_t k[QPT];   // bit j of k[h]: start 64 (QPT t + h) + j is kept
        uint32_t p[QPT];   // kept starts of the tile before that qword
        uint32_t wave_end; // ... before the first qword of the next wavefront
        uint32_t total;    // kept starts of the tile
    };
    // flag stream (between stage and resolve of the tile ahead) and, in the same space, the per-wavefront lists of kept starts
    // of the tile being emitted: the flag stream is dead by then (the scan's barrier lies between its last reader and the first
    // list entry, the barrier that opens the next front between the last list reader and the next flag)
    constexpr uint32_t LIST_BYTES = UMODE == UMODE_COUNT ? 0u : (uint32_t)WAVES * ULSTRIDE * 2u;
    constexpr uint32_t AMB_ALLOC = AMB_QWORDS * 8u > LIST_BYTES ? AMB_QWORDS : (LIST_BYTES + 7u) / 8u;
    __shared__ uint64_t amb[AMB_ALLOC];
    uint16_t *const kept = reinterpret_cast<uint16_t *>(amb);
    __shared__ uint64_t s_tile, s_base;
    __shared__ uint32_t s_wave_total[WAVES];
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += BLOCK) lut[i] = ascii_entry(a.ascii_table, i);
    }
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, 2);
    const uint32_t T = a.tile_starts;
    constexpr bool WIDE = N == 0;                        // kmers of more than four words: their width is a.n_words
    constexpr int NW = WIDE ? 1 : N;                     // (register arrays of the fixed-width paths)
    const uint32_t n_words = WIDE ? a.n_words : (uint32_t)N;
    uint64_t acc = 0;  // COUNT: kept starts; XOR: fold of the head words
    uint64_t lattice = ~0ull;  // bit j set iff j % stride == 0 (stride < 64; a larger stride has one lattice start per qword at most)
    if (a.stride > 1) {
        lattice = 0;
        for (uint32_t j = 0; j < 64u; j += a.stride) lattice |= 1ull << j;
    }

    struct Geom {
        uint64_t m0, w0;
        uint32_t mt, b0, nw, nq, kbit0;
    };
    auto geometry = [&](uint64_t tile) {
        Geom g;
        g.m0 = tile * T;
        const uint64_t left = a.n_cand - g.m0;
        g.mt = left < T ? (uint32_t)left : T;
        const uint64_t bit0 = a.first_bit + g.m0 * SRC_BITS;
        g.w0 = bit0 >> 6;
        g.b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(g.mt - 1) + k) * SRC_BITS;
        g.nw = (uint32_t)(((end_bit + 63) >> 6) - g.w0);
        g.nq = (g.mt + 63u) >> 6;
        g.kbit0 = g.nw * (128u / SRC_BITS) - 2u * (g.b0 + k);  // the kmer of tile start r sits at stream bit kbit0 - 2r
        return g;
    };
#ifdef KMERS_STAMPS
    uint64_t ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define USTAMP(i) ts[i] = __builtin_amdgcn_s_memrealtime()
#else
#define USTAMP(i)
#endif

    // ---- front of a tile: stage its source words once, resolve every candidate start, publish the AGGREGATE -----------
    auto front = [&](uint64_t tile, uint32_t buf) {
        uint64_t *const lds = lds2[buf];
        TileRegs tr;
        const Geom g = geometry(tile);
        const uint32_t nw = g.nw, b0 = g.b0, mt = g.mt, nq = g.nq;
        const uint64_t w0 = g.w0, m0 = g.m0;
        const uint32_t tile_rem = a.stride > 1 ? (uint32_t)(m0 % a.stride) : 0u;  // wave-uniform
        block_sync();  // the readers of this buffer (the tile before last), of the flag stream and of the list are done
        USTAMP(1);
        // stage: source words -> 2-bit codes (in kmer order) + one ambiguity flag per symbol
        // (all of a lane's loads of a batch are in flight before the first is used -- a 32768-start tile of a 4-bit source is
        // 2050 words, nine per lane -- so that a tile pays one memory latency; byte sources take two such batches)
        constexpr uint32_t PRE = 9;
        for (uint32_t wbase = 0; wbase < nw; wbase += PRE * BLOCK) {
            uint64_t xs[PRE];
#pragma unroll
            for (uint32_t j = 0; j < PRE; ++j) {
                const uint32_t wi = wbase + tid + j * BLOCK;
                xs[j] = wi < nw ? a.src[w0 + wi] : 0;
            }
#pragma unroll
            for (uint32_t j = 0; j < PRE; ++j) {
                const uint32_t wi = wbase + tid + j * BLOCK;
                const uint64_t x = xs[j];
                if (wi < nw) {
                    if constexpr (SRC_BITS == 8) {
                        uint32_t codes = 0, flags = 0;
                        uint64_t f = 0;
#pragma unroll
                        for (int b = 0; b < 8; ++b) {
                            uint32_t v = lut[(x >> (8 * b)) & 0xffu];
                            codes |= (v & 3u) << (2 * b);
                            flags |= (v >= 0xf0u ? 1u : 0u) << b;             // 0xf0: ambiguous -> skip the window
                            f |= (uint64_t)(v == 0xffu ? 1u : 0u) << (8 * b);  // 0xff: not a nucleotide -> throw
                        }
                        reinterpret_cast<uint16_t *>(lds)[nw - 1u - wi] = (uint16_t)(rev2_32(codes) >> 16);
                        reinterpret_cast<uint8_t *>(amb)[wi] = (uint8_t)flags;
                        if (f) report_bad_symbols<8, true>(a.err_slot, a.first_bit, a.n_bases, 1u, k, w0 + wi, f, x, a.index_origin);
                    } else if constexpr (SRC_BITS == 4) {
                        uint64_t bad;
                        uint32_t c = pack_4to2(x, bad);
                        reinterpret_cast<uint32_t *>(lds)[nw - 1u - wi] = rev2_32(c);
                        reinterpret_cast<uint16_t *>(amb)[wi] = (uint16_t)bad_bits16(bad);
                    } else {
                        lds[nw - 1u - wi] = rev2(x);
                        reinterpret_cast<uint32_t *>(amb)[wi] = 0u;  // 32 symbols per 2-bit word, none ambiguous
                    }
                }
            }
        }
        // (flag and code bits past the staged words only ever reach starts >= mt, which are masked out below)
        USTAMP(2);
        block_sync();
        USTAMP(3);
        // resolve: thread t owns QPT consecutive qwords of the keep mask (64 starts each)
        uint32_t c2[QPT];
        uint32_t c = 0;
#pragma unroll
        for (uint32_t h = 0; h < QPT; ++h) {
            const uint32_t q = QPT * tid + h;
            uint64_t keep = 0;
            if (q < nq) {
                keep = keep_qword(amb, 64u * q + b0, k);
                const uint32_t valid = mt - 64u * q;  // starts of this qword that exist
                if (valid < 64u) keep &= (1ull << valid) - 1ull;
                if (a.stride > 1) {                   // keep only starts with (m0 + 64q + j) % stride == 0
                    const uint32_t rem = (tile_rem + (64u * q) % a.stride) % a.stride;   // (m0 + 64q) % stride
                    const uint32_t first = rem ? a.stride - rem : 0u;                    // first lattice start of the qword, then every stride-th
                    keep = first < 64u ? keep & (lattice << first) : 0;
                }
            }
            tr.k[h] = keep;
            c2[h] = (uint32_t)__popcll(keep);
            c += c2[h];
        }
        uint32_t incl = c;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(incl, d, 64);
            if ((int)lane >= d) incl += y;
        }
        if (lane == 63) s_wave_total[wave] = incl;
        block_sync();
        uint32_t before = 0, tile_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
            const uint32_t wt = s_wave_total[w];
            if (w < wave) before += wt;
            tile_total += wt;
        }
        uint32_t running = before + incl - c;
#pragma unroll
        for (uint32_t h = 0; h < QPT; ++h) {
            tr.p[h] = running;
            running += c2[h];
        }
        tr.wave_end = before + (uint32_t)__shfl(incl, 63, 64);
        tr.total = tile_total;
        if (tid == 0) {
            if constexpr (UMODE == UMODE_COUNT) acc += tile_total;
            // the aggregate is out as early as it can be: the tiles behind this one wait for nothing else of it
#ifdef KMERS_TEST_ABORT
            if (tile != 1)
#endif
            if constexpr (EMIT) desc_store(a.desc + tile, (tile == 0 ? DESC_PREFIX : DESC_AGGREGATE) | (uint64_t)tile_total);
        }
        USTAMP(4);
        return tr;
    };

    // ---- back of a tile: its exclusive prefix (look-back), then list and emit its kept starts --------------------------
    auto back = [&](uint64_t tile, uint32_t buf, const TileRegs &tr) {
        const uint64_t *const lds = lds2[buf];
        const Geom g = geometry(tile);
        const uint32_t mt = g.mt, kbit0 = g.kbit0;
        const uint64_t m0 = g.m0;
        if constexpr (EMIT) {
            // (one wavefront looks back and the others wait at the barrier behind it: every wavefront looking back for itself --
            // no barrier, four times the descriptor polls -- measured 7-9 % slower, profiles/r03_tuning.md section 4)
            if (wave == 0) {
                // Sum the descriptors of the preceding tiles, nearest first, until one holds an inclusive PREFIX.  Every step
                // reads LOOKBACK x 64 descriptors with all loads in flight together; a tile that has not published yet is
                // polled alone (re-reading whole windows made 1280 wavefronts hammer the same few cache lines).
                uint64_t excl = 0;
                if (tile != 0) {
                    const uint32_t tile_total = tr.total;
                    long long pos = (long long)tile - 1;  // the nearest predecessor not yet accounted for
                    uint64_t part = 0;                    // this lane's share of the sum
                    bool found = false;
                    while (!found) {
                        uint64_t e[LOOKBACK];
#pragma unroll
                        for (int j = 0; j < LOOKBACK; ++j) {  // chunk j: the 64 tiles pos - 64j - lane; before tile 0: prefix 0
                            const long long idx = pos - (long long)(lane + 64u * (uint32_t)j);
                            e[j] = idx >= 0 ? desc_load(a.desc + idx) : DESC_PREFIX;
                        }
#pragma unroll
                        for (int j = 0; j < LOOKBACK; ++j) {
                            if (!found) {
                                const long long idx = pos - (long long)(lane + 64u * (uint32_t)j);
                                uint32_t fp, spins = 0;
                                for (;;) {
                                    const uint32_t st = (uint32_t)(e[j] >> 62);
                                    const uint64_t bp = __ballot(st == 2u);
                                    fp = bp ? (uint32_t)__builtin_ctzll(bp) : 64u;      // the nearest prefix of this chunk
                                    const bool wait = st == 0u && lane < fp;             // a nearer tile has not published yet
                                    if (__ballot(wait) == 0) break;
                                    __builtin_amdgcn_s_sleep(8);
                                    if ((++spins % SPIN_CHECK) == 0) {                   // bounded: see SPIN_LIMIT
                                        const bool raised = desc_load(a.abort_flag) != 0;
                                        if (raised || spins >= SPIN_LIMIT) {
                                            if (!raised && lane == 0) desc_store(a.abort_flag, 1ull);
                                            fp = 0;  // give up: the result of this call is discarded by the host
                                            break;
                                        }
                                    }
                                    if (wait) e[j] = desc_load(a.desc + idx);            // only the missing ones are read again
                                }
                                if (lane <= fp) part += e[j] & DESC_VALUE;  // aggregates up to and including the prefix
                                found = fp < 64u;
                            }
                        }
                        pos -= 64 * LOOKBACK;
                    }
                    excl = wave_sum64(part);
#ifdef KMERS_TEST_ABORT
                    if (tile != 1)  // (nor its prefix: tile 1 stays silent)
#endif
                    if (lane == 0) desc_store(a.desc + tile, DESC_PREFIX | ((excl + (uint64_t)tile_total) & DESC_VALUE));
                }
                if (lane == 0) s_base = excl;
            }
            USTAMP(5);
            block_sync();
            USTAMP(6);
        }
        const uint64_t base = EMIT ? s_base : 0;

        // every wavefront takes the CONTIGUOUS quarter of the tile whose keep mask its own lanes resolved, so that its stores
        // sweep one contiguous region of each output array and the mask never leaves the registers
        uint16_t *mine = kept + wave * ULSTRIDE;
        const uint32_t qw0 = wave * WQ;                                       // the wavefront's first qword
        const uint32_t wave_end = (qw0 + WQ) * 64u < mt ? (qw0 + WQ) * 64u : mt;
        // kept starts of the tile before qword qw0 + qrel (qrel <= WQ, the same in every lane)
        auto prefix_at = [&](uint32_t qrel) -> uint32_t {
            if (qrel >= WQ) return tr.wave_end;
            uint32_t v = 0;
#pragma unroll
            for (uint32_t h = 0; h < QPT; ++h) {
                const uint32_t x = (uint32_t)__shfl(tr.p[h], (int)(qrel / QPT), 64);
                if (qrel % QPT == h) v = x;
            }
            return v;
        };
        bool framed = false;
        if constexpr (EMIT && !WIDE) framed = a.vec16 && !a.tuples;
        if constexpr (EMIT && !WIDE) {
            if (framed) {
                // ---- the framed path (separate 16-byte aligned arrays, kmers of one to four words) -------------------------
                const uint32_t qbase = qw0 * 64u;                       // tile-relative start of the wavefront's quarter
                const uint64_t origin = m0 + 1 + a.index_origin;        // start of candidate r is origin + r
                uint32_t list_n = 0;                                    // listed, not yet emitted (quarter-relative starts)
                uint64_t list_pos = base + prefix_at(0);                // output index of the first element not yet emitted
                // elements with output indexes [lo, hi) -> memory, frame by frame; start_of(j) = tile-relative start of element lo + j
                auto emit_range = [&](uint64_t lo, uint64_t hi, auto start_of) {
                    // (elements at or beyond the capacity are not stored: the host reports KMERS_E_CAPACITY with the count needed)
                    const uint64_t room = a.capacity > lo ? a.capacity - lo : 0;
                    const uint32_t n = (uint32_t)(hi - lo < room ? hi - lo : room);
                    const uint32_t head = (uint32_t)lo & (UFRAME - 1u);  // where `lo` lies in its frame
                    for (uint32_t f = 0; f < head + n; f += UFRAME) {
                        const uint32_t j0 = f + 2u * lane - head, j1 = j0 + 1u;  // element numbers relative to lo (wrapped if before it)
                        const bool v0 = j0 < n, v1 = j1 < n;
                        if (!(v0 || v1)) continue;
                        uint32_t ra = 0, rb = 0;
                        uint64_t fa[NW], fb[NW];
                        if (v0) {
                            ra = start_of(j0);
                            cut_fw<NW>(lds, kbit0 - 2u * ra, mask, fa);
                        }
                        if (v1) {
                            rb = start_of(j1);
                            cut_fw<NW>(lds, kbit0 - 2u * rb, mask, fb);
                        }
                        const uint64_t i1 = lo + j1, i0 = i1 - 1u;  // (i1 is the one that always exists: j1 = 0 when `lo` is odd)
                        if (v0 && v1) {
                            if (a.out_kmers) {
                                if constexpr (N == 1) {
                                    *reinterpret_cast<ulonglong2 *>(a.out_kmers + i0) = make_ulonglong2(fa[0], fb[0]);
                                } else {
                                    store_kmer<NW>(a.out_kmers, i0, fa);
                                    store_kmer<NW>(a.out_kmers, i1, fb);
                                }
                            }
                            if (a.out_starts) *reinterpret_cast<ulonglong2 *>(a.out_starts + i0) = make_ulonglong2(origin + ra, origin + rb);
                        } else {  // the first element of an odd range / the last of one that ends on an even index
                            const uint64_t i = v0 ? i0 : i1;
                            if (a.out_kmers) {
#pragma unroll
                                for (int wd = 0; wd < NW; ++wd) a.out_kmers[i * NW + wd] = v0 ? fa[wd] : fb[wd];
                            }
                            if (a.out_starts) a.out_starts[i] = (long long)(origin + (v0 ? ra : rb));
                        }
                    }
                };
                auto listed = [&](uint32_t j) { return qbase + (uint32_t)mine[j]; };
                uint32_t n_round = 0;
                for (uint32_t r_begin = qbase; r_begin < wave_end; r_begin += n_round) {
                    const uint32_t q0 = r_begin >> 6;
                    const uint32_t round_off = prefix_at(q0 - qw0);
                    // a long round if its kept starts fit the list (or nothing at all is dropped), else 1024 starts
                    n_round = wave_end - r_begin < ULONG ? wave_end - r_begin : ULONG;
                    uint32_t q1 = (r_begin + n_round + 63u) >> 6;
                    uint32_t cnt = prefix_at(q1 - qw0) - round_off;
                    uint32_t usl = ULONG / 64u;
                    if (n_round > UROUND && cnt > ULIST && cnt != n_round) {
                        n_round = UROUND;
                        q1 = q0 + UROUND / 64u;
                        cnt = prefix_at(q1 - qw0) - round_off;
                        usl = USLICE;
                    } else if (n_round <= UROUND) {
                        usl = USLICE;
                    }
                    if (cnt == 0) continue;
                    if (cnt == n_round) {
                        // a round with nothing dropped (real sequence outside its N blocks) needs no list: element j starts at
                        // r_begin + j.  Whatever is still listed goes first (output order).
                        if (list_n) {
                            emit_range(list_pos, list_pos + list_n, listed);
                            list_pos += list_n;
                            list_n = 0;
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        emit_range(list_pos, list_pos + cnt, [&](uint32_t j) { return r_begin + j; });
                        list_pos += cnt;
                        continue;
                    }
                    // this lane's qword of the mask and the kept starts before it, from the lane that holds them
                    const uint32_t sl = (lane * usl) & 63u;
                    uint32_t qrel = q0 - qw0 + ((lane * usl) >> 6);
                    if (qrel >= WQ) qrel = WQ - 1u;  // (a lane past the end of the round: its slice is empty, any qword will do)
                    uint64_t km = 0;
                    uint32_t km_before = 0;
#pragma unroll
                    for (uint32_t h = 0; h < QPT; ++h) {
                        const uint64_t xk = __shfl(tr.k[h], (int)(qrel / QPT), 64);
                        const uint32_t xp = (uint32_t)__shfl(tr.p[h], (int)(qrel / QPT), 64);
                        if (qrel % QPT == h) {
                            km = xk;
                            km_before = xp;
                        }
                    }
                    const uint32_t mine_n = lane * usl < n_round ? (n_round - lane * usl < usl ? n_round - lane * usl : usl) : 0u;
                    const uint64_t keep16 = (km >> sl) & (mine_n >= 64u ? ~0ull : ((1ull << mine_n) - 1ull));
                    // list the round's kept starts behind the carried ones, in order, relative to the quarter
                    uint32_t o = list_n + km_before - round_off + (uint32_t)__popcll(km & ((1ull << sl) - 1ull));
                    const uint32_t s0 = r_begin - qbase + lane * usl;
                    uint32_t half_lo = (uint32_t)keep16, half_hi = (uint32_t)(keep16 >> 32);
                    while (half_lo) {
                        mine[o++] = (uint16_t)(s0 + (uint32_t)__builtin_ctz(half_lo));
                        half_lo &= half_lo - 1u;
                    }
                    while (half_hi) {
                        mine[o++] = (uint16_t)(s0 + 32u + (uint32_t)__builtin_ctz(half_hi));
                        half_hi &= half_hi - 1u;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t total = list_n + cnt;
                    const uint64_t end_emit = (list_pos + total) & ~(uint64_t)(UFRAME - 1u);  // the last whole frame's end
                    if (end_emit > list_pos) {
                        const uint32_t E = (uint32_t)(end_emit - list_pos), left = total - E;  // left < UFRAME
                        emit_range(list_pos, end_emit, listed);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        // what is left moves to the head of the list: every lane reads before any lane writes
                        uint16_t x0 = 0, x1 = 0;
                        if (lane < left) x0 = mine[E + lane];
                        if (lane + 64u < left) x1 = mine[E + lane + 64u];
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (lane < left) mine[lane] = x0;
                        if (lane + 64u < left) mine[lane + 64u] = x1;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        list_pos = end_emit;
                        list_n = left;
                    } else {
                        list_n = total;
                    }
                }
                if (list_n) emit_range(list_pos, list_pos + list_n, listed);  // the tail of the quarter: one partial frame
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next tile
            }
        }
        // ---- the generic path: one element per lane (Tuple{Kmer,Int} elements, unaligned outputs, kmers of run-time width, and
        // the XOR reducer, which stores nothing)
        uint32_t n_round = 0;
        for (uint32_t r_begin = qw0 * 64u; r_begin < wave_end && !framed; r_begin += n_round) {
            const uint32_t q0 = r_begin >> 6;
            const uint32_t round_off = prefix_at(q0 - qw0);
            // a long round if its kept starts fit the list (or nothing at all is dropped), else 1024 starts
            n_round = wave_end - r_begin < ULONG ? wave_end - r_begin : ULONG;
            uint32_t q1 = (r_begin + n_round + 63u) >> 6;
            uint32_t cnt = prefix_at(q1 - qw0) - round_off;                // kept starts of this round
            uint32_t usl = ULONG / 64u;                                    // consecutive starts per lane
            uint64_t pos = base + round_off;                               // output index of the round's first
            if (n_round > UROUND && cnt > ULIST) {
                n_round = UROUND;
                q1 = q0 + UROUND / 64u;
                cnt = prefix_at(q1 - qw0) - round_off;
                usl = USLICE;
            } else if (n_round <= UROUND) {
                usl = USLICE;
            }
            // this lane's qword of the mask and the kept starts before it, from the lane that holds them
            const uint32_t sl = (lane * usl) & 63u;
            uint32_t qrel = q0 - qw0 + ((lane * usl) >> 6);
            if (qrel >= WQ) qrel = WQ - 1u;  // (a lane past the end of the round: its slice is empty, any qword will do)
            uint64_t km = 0;
            uint32_t km_before = 0;
#pragma unroll
            for (uint32_t h = 0; h < QPT; ++h) {
                const uint64_t xk = __shfl(tr.k[h], (int)(qrel / QPT), 64);
                const uint32_t xp = (uint32_t)__shfl(tr.p[h], (int)(qrel / QPT), 64);
                if (qrel % QPT == h) {
                    km = xk;
                    km_before = xp;
                }
            }
            // this lane's slice of the keep mask: `usl` consecutive starts, cut at the end of the round (the starts behind it
            // belong to the next wavefront's chunk)
            const uint32_t mine_n = lane * usl < n_round ? (n_round - lane * usl < usl ? n_round - lane * usl : usl) : 0u;
            uint64_t keep16 = (km >> sl) & (mine_n >= 64u ? ~0ull : ((1ull << mine_n) - 1ull));
            if (cnt == 0) continue;
            const uint64_t origin = m0 + 1 + a.index_origin;                     // start of candidate r is origin + r
            // list the kept starts of the round in LDS, in order: this lane's slice of consecutive starts begins at list index o
            uint32_t o = km_before - round_off + (uint32_t)__popcll(km & ((1ull << sl) - 1ull));
            const uint32_t s0 = lane * usl;  // round-relative index of the slice's first start
            // (half by half: the 64-bit form of this loop is twelve instructions per listed start, a 32-bit half seven)
            uint32_t half_lo = (uint32_t)keep16, half_hi = (uint32_t)(keep16 >> 32);
            while (half_lo) {
                mine[o++] = (uint16_t)(s0 + (uint32_t)__builtin_ctz(half_lo));
                half_lo &= half_lo - 1u;
            }
            while (half_hi) {
                mine[o++] = (uint16_t)(s0 + 32u + (uint32_t)__builtin_ctz(half_hi));
                half_hi &= half_hi - 1u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // one listed element: window cut + stores (also the generic path: tuples, unaligned outputs, the XOR reducer)
            auto emit_one = [&](uint32_t i) {
                const uint32_t r = r_begin + (uint32_t)mine[i];
                if constexpr (WIDE) {
                    // run-time width: word n_words-1-j of the kmer = stream bits [o + 64j, +64), straight to memory
                    const uint32_t o = kbit0 - 2u * r, q = o >> 6, sh = o & 63u;
                    const uint64_t o2 = pos + i;
                    uint64_t lo = lds[q], head = 0;
                    for (uint32_t j = 0; j < n_words; ++j) {
                        const uint64_t hi = lds[q + j + 1u];
                        uint64_t w = funnel64(lo, hi, sh);
                        lo = hi;
                        if (j + 1u == n_words) head = w &= mask;
                        if constexpr (UMODE != UMODE_XOR) {
                            if (o2 < a.capacity && a.out_kmers) a.out_kmers[o2 * (n_words + (a.tuples ? 1u : 0u)) + (n_words - 1u - j)] = w;
                        }
                    }
                    if constexpr (UMODE == UMODE_XOR) {
                        acc ^= head;
                    } else if (o2 < a.capacity) {
                        if (a.tuples) a.out_kmers[o2 * (n_words + 1u) + n_words] = origin + r;
                        else if (a.out_starts) a.out_starts[o2] = (long long)(origin + r);
                    }
                    return;
                }
                uint64_t fw[NW];
                cut_fw<NW>(lds, kbit0 - 2u * r, mask, fw);
                if constexpr (UMODE == UMODE_XOR) {
                    acc ^= fw[0];
                } else {
                    const uint64_t o2 = pos + i;
                    if (o2 < a.capacity) {
                        if (a.tuples) {  // Tuple{Kmer,Int}: eltype of UnambiguousKmers (UnambiguousKmers.jl:39-41)
                            if (N == 1 && a.vec16) {  // one 16-byte store per element
                                *reinterpret_cast<ulonglong2 *>(a.out_kmers + 2u * o2) = make_ulonglong2(fw[0], origin + r);
                            } else {
#pragma unroll
                                for (int wd = 0; wd < NW; ++wd) a.out_kmers[o2 * (NW + 1) + wd] = fw[wd];
                                a.out_kmers[o2 * (NW + 1) + NW] = origin + r;
                            }
                        } else {
                            if (a.out_kmers) {
#pragma unroll
                                for (int wd = 0; wd < NW; ++wd) a.out_kmers[o2 * NW + wd] = fw[wd];
                            }
                            if (a.out_starts) a.out_starts[o2] = (long long)(origin + r);
                        }
                    }
                }
            };
            for (uint32_t i = lane; i < cnt; i += 64u) emit_one(i);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next round
        }
#ifdef KMERS_STAMPS
        USTAMP(7);  // this wavefront's stores issued
        if (a.stamps && (tile & 15u) == 0 && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint64_t *o = a.stamps + ((tile >> 4) * WAVES + wave) * 10;
            for (int i = 0; i < 8; ++i) o[i] = ts[i];
            o[8] = __builtin_amdgcn_s_memrealtime();  // ... and drained
            o[9] = tile;
        }
#endif
    };

    if constexpr (!EMIT) {
        // COUNT / XOR: a persistent grid strides over the tiles; nothing is placed, so no descriptors
        for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
            const TileRegs tr = front(tile, 0);
            if constexpr (UMODE == UMODE_XOR) {
                block_sync();
                back(tile, 0, tr);
            } else {
                (void)tr;
            }
        }
    } else {
        // EMIT: a persistent grid draws tiles from the ticket counter (tile ids in the order workgroups reach for them: a tile
        // only ever waits for lower ids, all held by running workgroups) and works one tile AHEAD on the front: tile n+1 is
        // staged, resolved and its aggregate published before tile n looks back and emits.  An aggregate is thus out a few
        // microseconds after its ticket was drawn, and by the time a tile looks back -- one whole front later -- its
        // predecessors have published theirs: the look-back finds what it needs at once (it waited 14 us per tile, with
        // three of the four wavefronts idle, when it ran right behind the tile's own resolve).
        auto draw = [&]() {
            block_sync();  // (everybody has read the previous s_tile)
            if (tid == 0) s_tile = atomicAdd(a.ticket, 1ull);
            block_sync();
            return (uint64_t)s_tile;
        };
        uint64_t cur = draw();
        uint32_t buf = 0;
        TileRegs tr_cur{}, tr_nxt{};
        if (cur < a.n_tiles) tr_cur = front(cur, buf);
        while (cur < a.n_tiles) {
#ifdef KMERS_STAMPS
            ts[0] = __builtin_amdgcn_s_memrealtime();
#endif
            const uint64_t nxt = draw();
            if (nxt < a.n_tiles) tr_nxt = front(nxt, buf ^ 1u);
            block_sync();  // the codes of the tile are staged for every wavefront
            back(cur, buf, tr_cur);
            tr_cur = tr_nxt;
            cur = nxt;
            buf ^= 1u;
        }
    }
    if constexpr (UMODE == UMODE_COUNT) {
        if (tid == 0 && acc) atomicAdd(a.total, (unsigned long long)acc);
    }
    if constexpr (UMODE == UMODE_XOR) {
        for (int off = 32; off > 0; off >>= 1) acc ^= __shfl_xor(acc, off, 64);
        if (lane == 0 && acc) atomicXor(a.total, (unsigned long long)acc);
    }
}

}  // namespace kmers
