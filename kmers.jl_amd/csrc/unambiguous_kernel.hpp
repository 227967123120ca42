// unambiguous_kernel.hpp -- UnambiguousKmers (src/iterators/UnambiguousKmers.jl:59-148) in ONE pass over the source:
// every window of K unambiguous symbols together with its 1-based start index, in the reference's order.
//
// The reference walks the sequence with a `remaining` counter that every ambiguous symbol resets (:140-146).  Here a
// window is kept iff the K bits of a one-bit-per-symbol ambiguity stream (staged in LDS next to the 2-bit code stream)
// are all zero, which selects exactly the same windows.  The output position of a kept window is the number of kept
// windows before it -- a prefix sum over the whole sequence -- and it is resolved INSIDE the emitting kernel:
//
//   tile     a workgroup draws a ticket (tiles are numbered in the order workgroups reach for them), stages its source words
//            once, resolves all of its candidate starts bit-parallel (64 starts per lane: the ambiguity bits OR-ed with
//            themselves shifted by 1, 2, 4, ...), and publishes its kept count as an AGGREGATE in its tile descriptor;
//   look-back   one wavefront then sums the descriptors of the preceding tiles, nearest first, 256 per step, until it meets
//            one that already holds an inclusive PREFIX (decoupled look-back: it never waits for a predecessor's prefix,
//            only for aggregates, which every started tile publishes unconditionally -- no circular wait), publishes its
//            own inclusive prefix and hands the exclusive one to the workgroup;
//   emit     every wavefront lists the kept starts of up to 4096 candidates at a time in LDS (their order is the reference's)
//            and works the list off with every lane busy: window cut + contiguous 16-byte stores in aligned frames.  A stretch
//            with nothing dropped (real sequence outside its N blocks) skips the list.
//   pipeline the grid is persistent (three workgroups per CU).  A workgroup runs the first step of its NEXT tile before the last
//            two of the current one, so that an aggregate is out microseconds after its ticket; the ticket itself is drawn
//            while the tile before it is resolved.  Between front and back a tile's keep mask stays in the registers of the
//            lanes that resolved it.  (Source words of the tile after next in flight while a tile is emitted: built,
//            KMERS_UPREFETCH, and slower -- the kernel does not wait for its loads, profiles/r04_unamb.md.)
//
// Round 4 (profiles/r04_unamb.md).  The kernel without its stores runs in 0.35 ms per Gbase on the C5 lattice and 0.50 at K = 31
// (close to what its vector instructions cost the SIMDs: 2.24 cycles each for the simple two-operand integer ones, 4.1 for all
// others, tools/device_probes/valu_rates.hip), its stores alone take 0.30 / 0.60 ms at the 7.4 TB/s a persistent grid writes into two region
// classes, and together they take 0.55 / 0.86: a wavefront that meets a full store queue stalls with its arithmetic behind it.
// What this version changed: a lane owns the qwords lane, 64 + lane, ... of its wavefront's quarter and a round is
// exactly one qword per lane (no lane shuffles, half the rounds), the list is frame-aligned and worked off two frames per step
// with the next step's list words in flight, the ticket is drawn inside the previous front, the recoding of a 4-bit word takes 41
// instructions instead of 59 and the window test runs on 32-bit words with v_alignbit (27 % fewer vector instructions on the
// lattice, 13 % at K = 31) -- and fewer, longer-lived-per-tile workgroups (3 x 65536): +4 % on the lattice, +2.5 % at K = 31.
//
// A descriptor is ONE 64-bit word (status in the top two bits, count below) moved with relaxed agent-scope atomics, so
// no fence is needed: the word is the whole hand-off (MI355X_MICROARCH.md, inter-workgroup visibility, "8-B agent atomics
// both sides").  The source is read once; there is no count pass, no scan launch and no host round trip before the emit.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

// 65536-start tiles, three workgroups per CU (46.5 KiB of LDS each).  Round 3 ran 4 x 49152 (with its kernel 6 x 32768, 5 x 40960
// and 3 x 65536 ran the same, 7 x 28672, 8 x 24576 and 2 x 98304 lost: profiles/r03_tuning.md); with this round's kernel, on one
// box: 4 x 49152 K = 31 0.900-0.917 ms / lattice 0.565-0.577, 3 x 49152 0.870-0.885 / 0.558, 3 x 65536 0.842-0.852 / 0.542-0.551
// (round 3's kernel beside them: 0.860-0.866 / 0.565; profiles/r04_unamb.md) -- the kernel writes at the rate the device takes
// stores from a persistent grid, and that rate falls with the number of resident workgroups (tools/device_probes/store_pacing.hip)
#ifndef KMERS_UTILE_MAX
#define KMERS_UTILE_MAX 65536
#endif
constexpr uint32_t UTILE_MAX = KMERS_UTILE_MAX;  // candidate starts per tile, at most: a multiple of 64 x BLOCK (whole qwords per thread)
constexpr uint32_t UROUND = 1024;                // granularity of the tile length (unambiguous_api.hip)
// A ROUND is one keep-mask qword per lane: 4096 consecutive candidate starts of the wavefront's quarter of the tile.  Its kept
// starts are listed in LDS in one PASS if they fit the wavefront's list, else in two passes (lanes 0-31, 32-63) or in passes of
// 16 lanes (1024 starts: always fit).
// Round 5, built and NOT shipped (KMERS_UINTERLEAVE=1; profiles/r05_unamb.md): a wavefront keeps TWO lists and lists its NEXT pass
// between the stores of the frames of the current one, so that a store that finds the queue full would have had the listing in front
// of it instead of behind it (the first of the two designs of profiles/r04_unamb.md).  Same results (the 200
// geometries of the fuzz), and slower at every instalment size: K = 31 1.15-1.19 ms against 0.92 with the same list length and
// 0.86 shipped; by the stamps a wavefront's listing + frames take 93 k cycles interleaved where they take 62 k one after the
// other -- the stores do not leave room that listing could fill from inside the same wavefront.
#ifndef KMERS_UINTERLEAVE
#define KMERS_UINTERLEAVE 0
#endif
#ifndef KMERS_ULIST
#define KMERS_ULIST (KMERS_UINTERLEAVE ? 1024 : 1536)
#endif
#ifndef KMERS_ULSTEP
#define KMERS_ULSTEP 16  // entries per lane listed behind every two frames' stores
#endif
constexpr uint32_t ULIST = KMERS_ULIST;
constexpr uint32_t ULISTS = KMERS_UINTERLEAVE ? 2u : 1u;  // lists per wavefront
// The emitting path stores in FRAMES: 128 consecutive output elements, aligned to 128 in the OUTPUT index (lane l of a frame owns
// elements 2l and 2l + 1: one 16-byte store per lane and array, 1 KiB = eight whole 128-byte lines per wave store).  The list
// is kept frame-aligned too (slot s of the list holds the element with output index frame_base + s), so that a lane reads its
// two entries with one 32-bit load; what a pass lists beyond its last whole frame stays at the head of the list for the next one
// (fewer than UFRAME entries), so a wavefront issues partial store instructions only at the two ends of its quarter of the tile.
constexpr uint32_t UFRAME = 128;
constexpr uint32_t ULSTRIDE = ULIST + UFRAME;  // list entries per wavefront: a pass's kept starts behind the carried ones
static_assert(ULIST >= 1024 && ULIST % 2 == 0, "a pass of 16 lanes (1024 starts) may keep every one of its starts");
#ifndef KMERS_UCUT
#define KMERS_UCUT 0  // diagnostic builds only: 1 = stage, 2 = whole front, 3 = front + look-back, 4 = loads only, 5 = stage without loads, 6 = everything but the stores of whole frames
#endif
#ifndef KMERS_UPREFETCH
#define KMERS_UPREFETCH 0  // source words per lane of the tile after next that EMIT loads while it emits (registers: two VGPRs each)
#endif
#ifndef KMERS_USTAGGER
#define KMERS_USTAGGER 0  // start delay of a workgroup in units of s_sleep 127 (3.6 us) per quarter step
#endif
#ifndef KMERS_USTAGGER_HASH
#define KMERS_USTAGGER_HASH 0
#endif
#ifndef KMERS_UNAMB_WGS
#define KMERS_UNAMB_WGS 3
#endif
constexpr int UNAMB_EMIT_WGS = KMERS_UNAMB_WGS;  // workgroups per CU of the emitting mode
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
    uint32_t tile_phase;           // 1: tile_starts is not a multiple of the stride (the lattice begins at a different place in every tile)
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
// inclusive prefix sum over the 64 lanes of a wavefront, in the vector pipe (six v_add with DPP operands; a __shfl_up ladder is
// six LDS-crossbar round trips): shifts inside rows of 16 lanes, then the last lane of a row broadcast to the rows behind it
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);  // row_shr:4 (lanes 4-15 of a row)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);  // row_shr:8 (lanes 8-15)
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}
// a value that is the same in every lane, moved to scalar registers
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
// lane `l` (the same in every lane) of v
__device__ __forceinline__ uint32_t lane_value(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }

// bit j of the result: K unambiguous symbols begin at symbol (bit + j) of the flag stream (one bit per symbol, set =
// ambiguous).  The window of start j spans flag bits [j, j + K): up to 64 + 127 bits for 64 starts and K <= 128.
// (K > 65 only: shorter kmers take keep_qword32 below)
__device__ __forceinline__ uint64_t keep_qword(const uint64_t *amb, uint32_t bit, uint32_t k) {
    const uint32_t Q = bit >> 6, sft = bit & 63u;
    uint64_t lo = ~funnel64(amb[Q], amb[Q + 1], sft), mid = ~funnel64(amb[Q + 1], amb[Q + 2], sft);
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
// The same for K <= 32 ND - 63 on 32-bit words: x[i] = flag bits [bit + 32 i, + 32) (ND words cover the 63 + K bits behind the
// 64 starts), x |= x >> step with the steps 1, 2, 4, ... (wave-uniform), every word one v_alignbit_b32 + one v_or_b32; what is
// shifted into the last word from beyond it only ever reaches starts above the 64.  `bit` < 32 (the stream starts in its word).
template <int ND>
__device__ __forceinline__ uint64_t keep_qword32(const uint32_t *amb32, uint32_t q, uint32_t bit, uint32_t k) {
    uint32_t d[ND + 1], x[ND];
#pragma unroll
    for (int i = 0; i <= ND; ++i) d[i] = amb32[2u * q + (uint32_t)i];
#pragma unroll
    for (int i = 0; i < ND; ++i) x[i] = __builtin_amdgcn_alignbit(d[i + 1], d[i], bit);
    uint32_t have = 1;
    while (have < k) {
        const uint32_t step = have < k - have ? have : k - have;  // 1..32
        if (step == 32u) {  // (v_alignbit takes its shift modulo 32)
#pragma unroll
            for (int i = 0; i + 1 < ND; ++i) x[i] |= x[i + 1];
        } else {
#pragma unroll
            for (int i = 0; i + 1 < ND; ++i) x[i] |= __builtin_amdgcn_alignbit(x[i + 1], x[i], step);
            x[ND - 1] |= x[ND - 1] >> step;
        }
        have += step;
    }
    return ~(((uint64_t)x[1] << 32) | x[0]);
}

// This kernel only ever needs FORWARD kmers, so it stages the 2-bit codes of a tile in KMER order (cut_fw, stream_kernel.hpp).

template <int SRC_BITS, int N, int UMODE>
__global__ __launch_bounds__(BLOCK, UMODE == UMODE_EMIT ? UNAMB_EMIT_WGS : 4) void unambiguous_kernel(const UnambArgs a) {
    constexpr bool EMIT = UMODE == UMODE_EMIT;
    constexpr uint32_t NBUF = EMIT ? 2u : 1u;  // EMIT resolves tile n+1 before it emits tile n: two sets of tile state
    constexpr uint32_t STREAM_QWORDS = (UTILE_MAX + 128 + 64) / 32 + 4;   // 2-bit codes of the tile + its K-1 overlap
    constexpr uint32_t AMB_QWORDS = (UTILE_MAX + 128 + 64) / 64 + 6;
    constexpr uint32_t MAXQ = UTILE_MAX / 64;
    static_assert(BLOCK == 256 && MAXQ % (uint32_t)BLOCK == 0, "whole keep-mask qwords per thread");
    __shared__ uint64_t lds2[NBUF][STREAM_QWORDS];
    // The keep mask of a tile never goes to LDS: wavefront w owns the qwords [WQ w, WQ (w + 1)) of the tile (64 starts each), its
    // lane l resolves the qwords l, 64 + l, ... of them and keeps them in registers until the tile is emitted (one whole front
    // later).  A round of the emitting half is then exactly one qword per lane: nothing moves between lanes.
    constexpr uint32_t QPT = MAXQ / (uint32_t)BLOCK;  // qwords per thread = rounds per wavefront (four with 65536-start tiles)
    constexpr uint32_t WQ = 64u * QPT;                // qwords per wavefront
    struct TileRegs {
        uint64_t k[QPT];     // bit j of k[h]: start 64 (WQ wave + 64 h + lane) + j of the tile is kept
        uint32_t wave_base;  // kept starts of the tile before the wavefront's quarter (the same in every lane)
        uint32_t total;      // kept starts of the tile
    };
    // flag stream (between stage and resolve of the tile ahead) and, in the same space, the per-wavefront lists of kept starts
    // of the tile being emitted: the flag stream is dead by then (the barrier behind the resolve lies between its last reader and
    // the first list entry, the barrier that opens the next front between the last list reader and the next flag)
    constexpr uint32_t LIST_BYTES = UMODE == UMODE_COUNT ? 0u : (uint32_t)WAVES * ULISTS * ULSTRIDE * 2u;
    constexpr uint32_t AMB_ALLOC = AMB_QWORDS * 8u > LIST_BYTES ? AMB_QWORDS : (LIST_BYTES + 7u) / 8u;
    __shared__ uint64_t amb[AMB_ALLOC];
    uint16_t *const kept = reinterpret_cast<uint16_t *>(amb);
    __shared__ uint64_t s_tile, s_base;
    __shared__ uint32_t s_wave_total[WAVES];
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += BLOCK) lut[i] = ascii_entry(a.ascii_table, i);  // (visible behind the front's first barrier)
    }
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, 2);
    const uint32_t T = a.tile_starts;
    constexpr bool WIDE = N == 0;                        // kmers of more than four words: their width is a.n_words
    constexpr int NW = WIDE ? 1 : N;                     // (register arrays of the fixed-width paths)
    const uint32_t n_words = WIDE ? a.n_words : (uint32_t)N;
    uint64_t acc = 0;  // COUNT: kept starts; XOR: fold of the head words
    // Stride lattices: bit j of `lattice` set iff j % stride == 0 (stride < 64; a larger stride has one lattice start per qword at
    // most); ph[h] = (tile-relative first start of this lane's qword h) % stride, once per kernel: the per-tile phase is then two
    // conditional subtractions (the general modulo was 45 instructions per qword, four of them multiplications)
    uint64_t lattice = ~0ull;
    uint32_t ph[QPT];
#pragma unroll
    for (uint32_t h = 0; h < QPT; ++h) ph[h] = 0;
    if (a.stride > 1) {
        lattice = 0;
        for (uint32_t j = 0; j < 64u; j += a.stride) lattice |= 1ull << j;
#pragma unroll
        for (uint32_t h = 0; h < QPT; ++h) ph[h] = (64u * (wave * WQ + 64u * h + lane)) % a.stride;
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
    uint64_t tp[6] = {0, 0, 0, 0, 0, 0}, tp0 = 0;  // shader cycles of a tile's back spent listing / emitting / carrying / scanning / in partial frames
#define USTAMP(i) ts[i] = __builtin_amdgcn_s_memrealtime()
#define UPHASE_BEGIN() tp0 = __builtin_amdgcn_s_memtime()
#define UPHASE_END(i) tp[i] += __builtin_amdgcn_s_memtime() - tp0
#else
#define USTAMP(i)
#define UPHASE_BEGIN()
#define UPHASE_END(i)
#endif

    // ---- the source words of a tile: PRE per lane, all in flight together.  EMIT issues them one whole `back` ahead --------
    // (a 65536-start tile of a 4-bit source is 4098 words + its alignment: seventeen per lane; byte sources take a second batch)
    constexpr uint32_t PRE = ((UTILE_MAX + 128u + 64u) * (SRC_BITS == 8 ? 4u : (uint32_t)SRC_BITS) / 64u + 2u + (uint32_t)BLOCK - 1u) / (uint32_t)BLOCK;
    constexpr uint32_t AHEAD = KMERS_UPREFETCH < PRE ? KMERS_UPREFETCH : PRE;  // words per lane loaded one whole `back` ahead
    constexpr uint32_t XS = AHEAD ? AHEAD : 1u;
    auto load_words = [&](uint64_t tile, uint64_t (&xs)[XS]) {
        if constexpr (AHEAD > 0) {
            const Geom g = geometry(tile);
#pragma unroll
            for (uint32_t j = 0; j < AHEAD; ++j) {
                const uint32_t wi = tid + j * BLOCK;
                xs[j] = wi < g.nw ? a.src[g.w0 + wi] : 0;
            }
        }
    };
    // one source word -> 2-bit codes (in kmer order) + one ambiguity flag per symbol
    auto stage_one = [&](uint64_t *lds, uint32_t nw, uint64_t w0, uint32_t wi, uint64_t x) {
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
            reinterpret_cast<uint32_t *>(lds)[nw - 1u - wi] = recode4_kmer_order(x);
            reinterpret_cast<uint16_t *>(amb)[wi] = (uint16_t)ambiguous16(x);
        } else {
            lds[nw - 1u - wi] = rev2(x);
            reinterpret_cast<uint32_t *>(amb)[wi] = 0u;  // 32 symbols per 2-bit word, none ambiguous
        }
    };

    // ---- front of a tile: stage its source words (already in registers), resolve every candidate start, publish the AGGREGATE
    // `ticket_next`: EMIT draws the ticket after next inside this front -- thread 0 of the workgroup issues the atomic behind the
    // stage and hands the result over at the barrier behind the resolve, so that neither its round trip nor a barrier of its own
    // is paid for (2.4 us under the kernel's store traffic, profiles/r03_tuning.md)
    auto front = [&](uint64_t tile, uint32_t buf, const uint64_t (&xs)[XS], bool ticket_next) {
        uint64_t *const lds = lds2[buf];
        TileRegs tr;
        const Geom g = geometry(tile);
        const uint32_t nw = g.nw, b0 = g.b0, mt = g.mt, nq = g.nq;
        const uint64_t w0 = g.w0, m0 = g.m0;
        // (m0 % stride, wave-uniform: a 64-bit division by a run-time value -- a hundred instructions -- unless the host could make
        // the tile length a multiple of the stride, unambiguous_api.hip)
        const uint32_t tile_rem = a.tile_phase ? (uint32_t)(m0 % a.stride) : 0u;
        // the words that were not loaded ahead: the rest of the first batch (all of its loads in flight before the first is used, and
        // while the prefetched words are staged), then further batches (byte sources, very long kmers)
        constexpr uint32_t REST = PRE - AHEAD;
        uint64_t ys[REST ? REST : 1u];
#pragma unroll
        for (uint32_t j = 0; j < REST; ++j) {
            const uint32_t wi = tid + (AHEAD + j) * BLOCK;
#if KMERS_UCUT == 5  // phase accounting: no loads at all (the stage works on made-up words)
            ys[j] = (uint64_t)(wi + 1u) * 0x9E3779B97F4A7C15ull + tile;
#else
            ys[j] = wi < nw ? a.src[w0 + wi] : 0;
#endif
        }
        unsigned long long drawn = 0;
        if constexpr (EMIT) {
            // (behind the loads of this tile in the memory queue, so that waiting for them does not wait for it; in flight until
            // the barrier behind the resolve)
            if (ticket_next && tid == 0) drawn = atomicAdd(a.ticket, 1ull);
        }
        block_sync();  // the readers of this buffer (the tile before last), of the flag stream and of the list are done
        USTAMP(1);
#pragma unroll
        for (uint32_t j = 0; j < AHEAD; ++j) {
            const uint32_t wi = tid + j * BLOCK;
            if (wi < nw) stage_one(lds, nw, w0, wi, xs[j]);
        }
#if KMERS_UCUT == 4  // phase accounting: the loads only, nothing recoded
        {
            uint64_t fold = 0;
#pragma unroll
            for (uint32_t j = 0; j < REST; ++j) fold ^= ys[j];
            if (fold == 0x6b6d657273756374ull) a.desc[tid] = fold;
        }
#else
#pragma unroll
        for (uint32_t j = 0; j < REST; ++j) {
            const uint32_t wi = tid + (AHEAD + j) * BLOCK;
            if (wi < nw) stage_one(lds, nw, w0, wi, ys[j]);
        }
#endif
        for (uint32_t wbase = PRE * BLOCK; wbase < nw; wbase += PRE * BLOCK) {
            uint64_t zs[PRE];
#pragma unroll
            for (uint32_t j = 0; j < PRE; ++j) {
                const uint32_t wi = wbase + tid + j * BLOCK;
                zs[j] = wi < nw ? a.src[w0 + wi] : 0;
            }
#pragma unroll
            for (uint32_t j = 0; j < PRE; ++j) {
                const uint32_t wi = wbase + tid + j * BLOCK;
                if (wi < nw) stage_one(lds, nw, w0, wi, zs[j]);
            }
        }
        // (flag and code bits past the staged words only ever reach starts >= mt, which are masked out below)
        USTAMP(2);
        block_sync();
        USTAMP(3);
#if KMERS_UCUT == 1 || KMERS_UCUT == 4 || KMERS_UCUT == 5  // phase accounting (tools/unamb_account.sh): the kernel ends behind the stage; the read keeps the LDS stores alive
        if (a.capacity == 0x6b6d657273756374ull) a.desc[tid] = lds[tid] + amb[tid];
        tr = TileRegs{};
        if (tid == 0) s_tile = drawn;
        block_sync();
        return tr;
#endif
        // resolve: this lane's qwords of the keep mask (64 starts each)
        const uint32_t *const amb32 = reinterpret_cast<const uint32_t *>(amb);
        uint32_t c = 0;
#pragma unroll
        for (uint32_t h = 0; h < QPT; ++h) {
            const uint32_t q = wave * WQ + 64u * h + lane;
            uint64_t keep = 0;
            if (64u * (wave * WQ + 64u * h) < mt) {  // (wave-uniform: the round exists)
                if (q < nq) {
                    if (k <= 33u) keep = keep_qword32<3>(amb32, q, b0, k);
                    else if (k <= 65u) keep = keep_qword32<4>(amb32, q, b0, k);
                    else keep = keep_qword(amb, 64u * q + b0, k);
                }
                if (64u * (wave * WQ + 64u * h + 64u) > mt) {  // (wave-uniform: the round holds the end of the tile)
                    const uint32_t valid = q < nq ? mt - 64u * q : 0u;  // starts of this qword that exist
                    if (valid < 64u) keep &= (1ull << valid) - 1ull;
                }
                if (a.stride > 1) {  // keep only starts with (m0 + 64q + j) % stride == 0
                    uint32_t rem = ph[h] + tile_rem;                    // (m0 + 64q) % stride, before the reduction: < 2 stride
                    rem = rem - a.stride < rem ? rem - a.stride : rem;  // (unsigned: the difference wraps when rem < stride)
                    const uint32_t first = rem ? a.stride - rem : 0u;   // first lattice start of the qword, then every stride-th
                    keep = first < 64u ? keep & (lattice << first) : 0;
                }
            }
            tr.k[h] = keep;
            c += (uint32_t)__popcll(keep);
        }
        const uint32_t incl = wave_scan_incl(c);
        if (lane == 63) s_wave_total[wave] = incl;
        if constexpr (EMIT) {
            if (ticket_next && tid == 0) s_tile = drawn;
        }
        block_sync();
        uint32_t before = 0, tile_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < (uint32_t)WAVES; ++w) {
            const uint32_t wt = s_wave_total[w];
            if (w < wave) before += wt;
            tile_total += wt;
        }
        tr.wave_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);  // (the same in every lane: scalar registers)
        tr.total = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_total);
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
#if KMERS_UCUT == 1 || KMERS_UCUT == 2 || KMERS_UCUT == 4 || KMERS_UCUT == 5  // phase accounting: fronts only (nobody looks back, so nobody waits)
        if (a.capacity == 0x6b6d657273756374ull) a.desc[tid] = lds[tid] + tr.k[0] + tr.wave_base + tr.total;
        return;
#endif
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
        const uint64_t base = EMIT ? uniform64(s_base) : 0;
#if KMERS_UCUT == 3  // phase accounting: fronts and look-backs, nothing listed or stored
        if (a.capacity == 0x6b6d657273756374ull) a.desc[tid] = lds[tid] + tr.k[0] + tr.wave_base + base;
        return;
#endif

        // every wavefront takes the CONTIGUOUS quarter of the tile whose keep mask its own lanes resolved, so that its stores
        // sweep one contiguous region of each output array and the mask never leaves the registers
        // the wavefront's lists: the one being worked off begins at entry m_off, the one the next pass is listed into meanwhile
        // (interleaved emit) at o_off -- offsets into one LDS array, not pointers that change places: those went through scratch memory
        uint16_t *const lists = kept + wave * ULISTS * ULSTRIDE;
        uint32_t m_off = 0, o_off = (ULISTS - 1u) * ULSTRIDE;
#define mine (lists + m_off)
#define other (lists + o_off)
        const uint32_t qbase = wave * WQ * 64u;                 // tile-relative start of the wavefront's quarter
        const uint64_t origin = m0 + 1 + a.index_origin;        // start of candidate r is origin + r
        uint32_t roff = tr.wave_base;                           // kept starts of the tile before the current round
        bool framed = false;
        if constexpr (EMIT && !WIDE) framed = a.vec16 && !a.tuples;

        // state of the framed path: slot s of the wavefront's list holds the element with output index frame_base + s; the
        // listed, not yet emitted elements are the slots [head, head + list_n)
        uint64_t frame_base = (base + roff) & ~(uint64_t)(UFRAME - 1u);
        uint32_t head = (uint32_t)(base + roff) & (UFRAME - 1u), list_n = 0;

        // elements with output indexes [lo, hi) -> memory, frame by frame with per-lane bounds (partial frames, rounds with nothing
        // dropped); start_of(j) = tile-relative start of element lo + j
        auto emit_range = [&](uint64_t lo, uint64_t hi, auto start_of) {
            if constexpr (EMIT && !WIDE) {
                // (elements at or beyond the capacity are not stored: the host reports KMERS_E_CAPACITY with the count needed)
                const uint64_t room = a.capacity > lo ? a.capacity - lo : 0;
                const uint32_t n = (uint32_t)(hi - lo < room ? hi - lo : room);
                const uint32_t hd = (uint32_t)lo & (UFRAME - 1u);  // where `lo` lies in its frame
                UPHASE_BEGIN();
                for (uint32_t f = 0; f < hd + n; f += UFRAME) {
                    const uint32_t j0 = f + 2u * lane - hd, j1 = j0 + 1u;  // element numbers relative to lo (wrapped if before it)
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
                UPHASE_END(4);
            }
        };
        // The pending LISTING JOB of the interleaved emit: the kept starts of the next pass, still to be written to `other` (this
        // lane's slots from job_o on).  list_some works off up to n entries per lane -- the low halves of all lanes first, then the
        // high halves, exactly list_lanes' order and instructions, in instalments.
        // (one "current half" per lane and a wave-uniform switch to the high halves: written as two symmetric branches the
        // compiler merged them into one body over a two-element array in scratch memory, 3.6 times the kernel's time)
        uint32_t job_cur = 0, job_nxt = 0, job_o = 0, job_s0 = 0;
        auto list_some = [&](uint32_t n) {
            uint32_t i = n;  // (list_lanes' loop with a budget)
            while (job_cur != 0u && i != 0u) {
                other[job_o++] = (uint16_t)(job_s0 + (uint32_t)__builtin_ctz(job_cur));
                job_cur &= job_cur - 1u;
                --i;
            }
            if (__ballot(job_cur != 0u) == 0) {  // every lane is through with its low half: on to the high halves
                job_cur = job_nxt;
                job_nxt = 0;
                job_s0 += 32u;
            }
        };
        // whole frames of the list: slots [128 f0, 128 f1) -> memory, every lane two elements, no bounds (the caller checked the
        // capacity); a lane's two entries are one aligned 32-bit word of the list.  Behind the stores of every step: an instalment
        // of the pending listing job (nothing if there is none).
        auto emit_frames = [&](uint32_t f0, uint32_t f1) {
            if constexpr (EMIT && !WIDE) {
                const uint32_t *const pairs = reinterpret_cast<const uint32_t *>(mine);
                auto put = [&](uint64_t i0, uint32_t ra, uint32_t rb, const uint64_t (&fa)[NW], const uint64_t (&fb)[NW]) {
#if KMERS_UCUT == 6  // phase accounting: the whole kernel without the stores of its whole frames
                    if (fa[0] != 0x6b6d657273756374ull) return;
#endif
                    if (a.out_kmers) {
                        if constexpr (N == 1) {
                            *reinterpret_cast<ulonglong2 *>(a.out_kmers + i0) = make_ulonglong2(fa[0], fb[0]);
                        } else {
                            store_kmer<NW>(a.out_kmers, i0, fa);
                            store_kmer<NW>(a.out_kmers, i0 + 1u, fb);
                        }
                    }
                    if (a.out_starts) *reinterpret_cast<ulonglong2 *>(a.out_starts + i0) = make_ulonglong2(origin + ra, origin + rb);
                };
                uint32_t f = f0;
                if constexpr (N == 1) {
                    if (a.out_kmers && a.out_starts && f + 2u <= f1) {
                        // The usual case (one-word kmers, both arrays), two frames per step and software-pipelined: a wavefront of
                        // this kernel pays one LDS round trip per DEPENDENT step whatever the number of loads in it, and the faster
                        // an emitting wavefront gets its stores out, the fewer of them it takes to keep the CU's store path busy
                        // (profiles/r04_unamb.md).  The list words of the next step are in flight while the code words of this one
                        // are cut; all twelve code words of a step are read before the first is used.
                        const uint32_t *const codes = reinterpret_cast<const uint32_t *>(lds);
                        const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
                        uint32_t p0 = pairs[64u * f + lane], p1 = pairs[64u * f + 64u + lane];
                        for (; f + 2u <= f1; f += 2u) {
                            const uint32_t r[4] = {qbase + (p0 & 0xffffu), qbase + (p0 >> 16), qbase + (p1 & 0xffffu), qbase + (p1 >> 16)};
                            uint32_t w[4][3], o[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                o[e] = kbit0 - 2u * r[e];
                                const uint32_t *const d = codes + (o[e] >> 5);
                                w[e][0] = d[0];
                                w[e][1] = d[1];
                                w[e][2] = d[2];
                            }
                            if (f + 4u <= f1) {  // (wave-uniform)
                                p0 = pairs[64u * f + 128u + lane];
                                p1 = pairs[64u * f + 192u + lane];
                            }
                            uint32_t lo[4], hi[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                lo[e] = __builtin_amdgcn_alignbit(w[e][1], w[e][0], o[e] & 31u) & mlo;
                                hi[e] = __builtin_amdgcn_alignbit(w[e][2], w[e][1], o[e] & 31u) & mhi;
                            }
                            const uint64_t i0 = frame_base + UFRAME * f + 2u * lane;
#if KMERS_UCUT == 6  // phase accounting: the whole kernel without the stores of its whole frames
                            if ((lo[0] ^ lo[1] ^ lo[2] ^ lo[3] ^ hi[0] ^ hi[1]) != 0x6b6d6572u) continue;
#endif
                            *reinterpret_cast<uint4 *>(a.out_kmers + i0) = make_uint4(lo[0], hi[0], lo[1], hi[1]);
                            *reinterpret_cast<ulonglong2 *>(a.out_starts + i0) = make_ulonglong2(origin + r[0], origin + r[1]);
                            *reinterpret_cast<uint4 *>(a.out_kmers + i0 + UFRAME) = make_uint4(lo[2], hi[2], lo[3], hi[3]);
                            *reinterpret_cast<ulonglong2 *>(a.out_starts + i0 + UFRAME) = make_ulonglong2(origin + r[2], origin + r[3]);
                            if constexpr (KMERS_UINTERLEAVE) list_some(KMERS_ULSTEP);
                        }
                    }
                }
                for (; f < f1; ++f) {
                    const uint32_t pr = pairs[64u * f + lane];
                    const uint32_t ra = qbase + (pr & 0xffffu), rb = qbase + (pr >> 16);
                    uint64_t fa[NW], fb[NW];
                    cut_fw<NW>(lds, kbit0 - 2u * ra, mask, fa);
                    cut_fw<NW>(lds, kbit0 - 2u * rb, mask, fb);
                    put(frame_base + UFRAME * f + 2u * lane, ra, rb, fa, fb);
                    if constexpr (KMERS_UINTERLEAVE) list_some(KMERS_ULSTEP / 2);
                }
            }
        };
        auto listed = [&](uint32_t j) { return qbase + (uint32_t)mine[head + j]; };  // element j of the pending list
#define LIST_FENCE(order)                          \
    do {                                           \
        __builtin_amdgcn_fence(order, "wavefront"); \
        __builtin_amdgcn_wave_barrier();           \
    } while (0)
        // everything still listed -> memory (a partial frame at its end); afterwards the list is empty and begins at `next`
        auto flush_list = [&](uint64_t next) {
            if (list_n) {
                emit_range(frame_base + head, frame_base + head + list_n, listed);
                LIST_FENCE(__ATOMIC_ACQUIRE);  // the list is rewritten by the next pass
            }
            frame_base = next & ~(uint64_t)(UFRAME - 1u);
            head = (uint32_t)next & (UFRAME - 1u);
            list_n = 0;
        };
        // the kept starts of the lanes [la, lb) of a round -> list slots, in order; o = this lane's first slot, s0 the
        // quarter-relative start of its qword
        auto list_lanes = [&](uint64_t km, uint32_t la, uint32_t lb, uint32_t o, uint32_t s0) {
            uint32_t lo32 = (uint32_t)km, hi32 = (uint32_t)(km >> 32);
            if (lane < la || lane >= lb) lo32 = hi32 = 0;
            // (half by half: the 64-bit form of this loop is twelve instructions per listed start, a 32-bit half seven; the two
            // halves side by side in one loop: 12.5, profiles/r04_unamb.md)
            while (lo32) {
                mine[o++] = (uint16_t)(s0 + (uint32_t)__builtin_ctz(lo32));
                lo32 &= lo32 - 1u;
            }
            while (hi32) {
                mine[o++] = (uint16_t)(s0 + 32u + (uint32_t)__builtin_ctz(hi32));
                hi32 &= hi32 - 1u;
            }
        };
        // a pass of a round: the lanes [la, lb) -- all that are left if their kept starts fit the list, else up to the next
        // multiple of 32, else 16 lanes (1024 starts: always fit).  Returns lb; `cntp` = kept starts of the pass, `ea` = kept
        // starts of the round before lane la
        auto next_pass = [&](uint32_t la, uint32_t excl, uint32_t cnt, uint32_t &ea, uint32_t &cntp) {
            ea = lane_value(excl, la);
            uint32_t lb = 64u;
            cntp = cnt - ea;
            if (cntp > ULIST) {
                lb = (la & ~31u) + 32u;
                cntp = (lb < 64u ? lane_value(excl, lb) : cnt) - ea;
                if (cntp > ULIST) {
                    lb = la + 16u;
                    cntp = (lb < 64u ? lane_value(excl, lb) : cnt) - ea;
                }
            }
            return lb;
        };

        // (the masks of the rounds move down one place after every round, so that the loop always works on rot[0]: indexing the
        // registers with the loop counter would put the tile's masks into scratch memory)
        uint64_t rot[QPT];
#pragma unroll
        for (uint32_t hh = 0; hh < QPT; ++hh) rot[hh] = tr.k[hh];
        bool pre_listed = false;  // (interleaved emit) the pass about to begin is listed already
#pragma unroll 1
        for (uint32_t h = 0; h < QPT; ++h) {
            const uint32_t r_begin = qbase + 4096u * h;  // tile-relative first start of the round
            if (r_begin >= mt) break;
            const uint32_t n_round = mt - r_begin < 4096u ? mt - r_begin : 4096u;
            UPHASE_BEGIN();
            const uint64_t km = rot[0];
#pragma unroll
            for (uint32_t hh = 0; hh + 1 < QPT; ++hh) rot[hh] = rot[hh + 1];
            const uint32_t c = (uint32_t)__popcll(km);
            const uint32_t incl = wave_scan_incl(c), excl = incl - c;  // kept starts of the round before this lane's qword
            const uint32_t cnt = lane_value(incl, 63);
            UPHASE_END(3);
            if (cnt == 0) continue;
            const uint32_t s0 = 4096u * h + 64u * lane;  // quarter-relative first start of this lane's qword
            if (framed) {
                if (cnt == n_round) {
                    // a round with nothing dropped (real sequence outside its N blocks) needs no list: element j starts at
                    // r_begin + j.  Whatever is still listed goes first (output order).
                    flush_list(base + roff);
                    emit_range(base + roff, base + roff + cnt, [&](uint32_t j) { return r_begin + j; });
                    roff += cnt;
                    flush_list(base + roff);
                    continue;
                }
                for (uint32_t la = 0; la < 64u;) {
                    uint32_t ea, cntp;
                    const uint32_t lb = next_pass(la, excl, cnt, ea, cntp);
                    UPHASE_BEGIN();
                    if (!pre_listed) {  // (else: listed between the stores of the pass before)
                        list_lanes(km, la, lb, head + list_n + excl - ea, s0);
                        LIST_FENCE(__ATOMIC_RELEASE);
                    }
                    pre_listed = false;
                    UPHASE_END(0);
                    const uint32_t end = head + list_n + cntp;          // slots [head, end) are listed
                    uint32_t f0 = 0;
                    if (head && end >= UFRAME) {  // the first frame of the quarter begins in the middle: per-lane bounds
                        emit_range(frame_base + head, frame_base + UFRAME, listed);
                        f0 = 1;
                    }
                    const uint32_t f1 = end / UFRAME;                    // whole frames: [f0, f1)
                    if (f1 > f0 || f0) {
                        const uint32_t E = UFRAME * f1, left = end - E;  // left < UFRAME: what stays listed
                        const bool whole = f1 > f0 && frame_base + (uint64_t)UFRAME * f1 <= a.capacity;
                        // ---- the pass after this one, if its listing can ride along: the next lanes of this round, or the first
                        //      pass of the next round unless that round is empty, ends the tile or keeps everything (no list)
                        bool ride = false;
                        if constexpr (KMERS_UINTERLEAVE) {
                            uint64_t nkm = km;
                            uint32_t nla = lb, nexcl = excl, ncnt = cnt, ns0 = s0;
                            ride = whole;
                            if (lb >= 64u) {
                                ride = false;
                                const uint32_t nr_begin = r_begin + 4096u;
                                if (whole && h + 1u < QPT && nr_begin < mt) {
                                    nkm = rot[0];  // (the masks have moved down already: the next round's)
                                    const uint32_t nn_round = mt - nr_begin < 4096u ? mt - nr_begin : 4096u;
                                    const uint32_t nc = (uint32_t)__popcll(nkm), nincl = wave_scan_incl(nc);
                                    nexcl = nincl - nc;
                                    ncnt = lane_value(nincl, 63);
                                    nla = 0;
                                    ns0 = s0 + 4096u;
                                    ride = ncnt != 0u && ncnt != nn_round;
                                }
                            }
                            if (ride) {
                                uint32_t nea, ncntp;
                                const uint32_t nlb = next_pass(nla, nexcl, ncnt, nea, ncntp);
                                job_cur = (uint32_t)nkm;
                                job_nxt = (uint32_t)(nkm >> 32);
                                if (lane < nla || lane >= nlb) job_cur = job_nxt = 0;
                                job_o = left + nexcl - nea;  // (behind what this pass leaves: head = 0, list_n = left from here on)
                                job_s0 = ns0;
                            }
                        }
                        UPHASE_BEGIN();
                        if (ride) {
                            // what this pass leaves moves to the head of the OTHER list, then its frames go out with the next
                            // pass's listing between their stores; the lists change places
                            if (lane < left) other[lane] = mine[E + lane];
                            if (lane + 64u < left) other[lane + 64u] = mine[E + lane + 64u];
                            emit_frames(f0, f1);
                            list_some(64u);  // (whatever of the job the frames did not cover: the rest of one half,
                            list_some(64u);  // then all of the other)
                            LIST_FENCE(__ATOMIC_ACQ_REL);
                            const uint32_t t_off = m_off;
                            m_off = o_off;
                            o_off = t_off;
                            pre_listed = true;
                            UPHASE_END(1);
                        } else {
                            if (f1 > f0) {
                                if (whole) emit_frames(f0, f1);
                                else emit_range(frame_base + (uint64_t)UFRAME * f0, frame_base + (uint64_t)UFRAME * f1,
                                                [&](uint32_t j) { return qbase + (uint32_t)mine[UFRAME * f0 + j]; });
                            }
                            LIST_FENCE(__ATOMIC_ACQUIRE);
                            UPHASE_END(1);
                            UPHASE_BEGIN();
                            // what is left moves to the head of the list: every lane reads before any lane writes
                            uint16_t x0 = 0, x1 = 0;
                            if (lane < left) x0 = mine[E + lane];
                            if (lane + 64u < left) x1 = mine[E + lane + 64u];
                            LIST_FENCE(__ATOMIC_ACQ_REL);
                            if (lane < left) mine[lane] = x0;
                            if (lane + 64u < left) mine[lane + 64u] = x1;
                            LIST_FENCE(__ATOMIC_RELEASE);
                            UPHASE_END(2);
                        }
                        frame_base += E;
                        head = 0;
                        list_n = left;
                    } else {
                        list_n += cntp;
                    }
                    la = lb;
                }
                roff += cnt;
                continue;
            }
            // ---- the generic path: one element per lane (Tuple{Kmer,Int} elements, unaligned outputs, kmers of run-time width, and
            // the XOR reducer, which stores nothing)
            for (uint32_t la = 0; la < 64u;) {
                uint32_t ea, cntp;
                const uint32_t lb = next_pass(la, excl, cnt, ea, cntp);
                list_lanes(km, la, lb, excl - ea, s0);
                LIST_FENCE(__ATOMIC_RELEASE);
                const uint64_t pos = base + roff + ea;  // output index of the pass's first element
                // one listed element: window cut + stores
                auto emit_one = [&](uint32_t i) {
                    const uint32_t r = qbase + (uint32_t)mine[i];
                    if constexpr (WIDE) {
                        // run-time width: word n_words-1-j of the kmer = stream bits [o + 64j, +64), straight to memory
                        const uint32_t o = kbit0 - 2u * r, q = o >> 6, sh = o & 63u;
                        const uint64_t o2 = pos + i;
                        uint64_t lo = lds[q], head_word = 0;
                        for (uint32_t j = 0; j < n_words; ++j) {
                            const uint64_t hi = lds[q + j + 1u];
                            uint64_t w = funnel64(lo, hi, sh);
                            lo = hi;
                            if (j + 1u == n_words) head_word = w &= mask;
                            if constexpr (UMODE != UMODE_XOR) {
                                if (o2 < a.capacity && a.out_kmers) a.out_kmers[o2 * (n_words + (a.tuples ? 1u : 0u)) + (n_words - 1u - j)] = w;
                            }
                        }
                        if constexpr (UMODE == UMODE_XOR) {
                            acc ^= head_word;
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
                for (uint32_t i = lane; i < cntp; i += 64u) emit_one(i);
                LIST_FENCE(__ATOMIC_ACQUIRE);  // the list is rewritten by the next pass
                la = lb;
            }
            roff += cnt;
        }
        if (framed) flush_list(0);  // the tail of the quarter: one partial frame
#ifdef KMERS_STAMPS
        USTAMP(7);  // this wavefront's stores issued
        if (a.stamps && (tile & 15u) == 0 && lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            uint64_t *o = a.stamps + ((tile >> 4) * WAVES + wave) * 16;
            for (int i = 0; i < 8; ++i) o[i] = ts[i];
            o[8] = __builtin_amdgcn_s_memrealtime();  // ... and drained
            o[9] = tile;
            for (int i = 0; i < 6; ++i) o[10 + i] = tp[i];
        }
        for (int i = 0; i < 6; ++i) tp[i] = 0;
#endif
    };

#undef LIST_FENCE
#undef mine
#undef other

    if constexpr (!EMIT) {
        // COUNT / XOR: a persistent grid strides over the tiles; nothing is placed, so no descriptors
        for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
            uint64_t xs[XS];
            load_words(tile, xs);
            const TileRegs tr = front(tile, 0, xs, false);
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
        // three of the four wavefronts idle, when it ran right behind the tile's own resolve).  The tile after next (`nn`) has
        // its ticket drawn inside the front of `nxt` and its source words in flight while `cur` is emitted.  (A workgroup holds
        // three tickets; the smallest tile without an aggregate is either about to be resolved by a workgroup whose look-back --
        // for a smaller tile -- finds every aggregate it needs, or it is that workgroup's `nn`, one `back` of that kind away
        // from its front: still no circular wait.)
        uint64_t xs[XS];
#if KMERS_USTAGGER > 0
        // The workgroups of a launch start together and their tiles take about the same time: without this their fronts (no
        // stores) and their backs (nothing but stores) stay in step across the whole device, and the store path idles through
        // every front.  A quarter, a half, three quarters of a tile time late, before the first ticket is drawn (nobody waits for
        // a workgroup that holds no ticket).
        {
            const uint32_t step = (KMERS_USTAGGER_HASH ? (blockIdx.x * 2654435761u) >> 30 : (blockIdx.x >> 8)) & 3u;
            for (uint32_t i = 0; i < step * (uint32_t)KMERS_USTAGGER; ++i) __builtin_amdgcn_s_sleep(127);
        }
#endif
        block_sync();
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1ull);
        block_sync();
        uint64_t cur = uniform64(s_tile);
        uint32_t buf = 0;
        TileRegs tr_cur{}, tr_nxt{};
        uint64_t nxt = ~0ull;
        if (cur < a.n_tiles) {
            load_words(cur, xs);
            tr_cur = front(cur, buf, xs, true);  // (draws the second ticket on the way)
            nxt = uniform64(s_tile);
            if (nxt < a.n_tiles) load_words(nxt, xs);
        }
        while (cur < a.n_tiles) {
#ifdef KMERS_STAMPS
            ts[0] = __builtin_amdgcn_s_memrealtime();
#endif
            uint64_t nn = ~0ull;
            if (nxt < a.n_tiles) {
                tr_nxt = front(nxt, buf ^ 1u, xs, true);
                nn = uniform64(s_tile);
                if (nn < a.n_tiles) load_words(nn, xs);  // in flight while `cur` is emitted
            }
            block_sync();  // the codes of the tile are staged for every wavefront
            back(cur, buf, tr_cur);
            tr_cur = tr_nxt;
            cur = nxt;
            nxt = nn;
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
