// ragged_kernels.hpp -- the iterators over a BATCH of records (reads, contigs, FASTA records)
// packed in one pool of source words.  The reference iterates one sequence at a time
// (`for record in reader; ... CanonicalDNAMers{K}(sequence(record)) ...`, docs/src/minhash.md:31-35);
// on the GPU a call costs ~16 us, so short records are only worth it many at a time: one launch
// produces the elements of all records, concatenated in record order.
//
//   1. recode_kernel   (skipped for Copyable 2->2 / 4->4 pools): the pool's symbols -> a DST-bit
//      stream in HBM + one "cannot be encoded" flag per symbol (RecodingScheme,
//      src/construction.jl:75-100).  Elementwise, r B/base read, (DST+1)/8 B/base written.
//   2. the layout pass: elements per record, exclusive scan, and one descriptor per tile of output
//      elements (first record, record slots, the stretch of stream words its windows lie in).
//   3. ragged_kernel: one tile (1024 .. 8192 output elements, chosen per call) per workgroup.  The
//      tile's slice of (element offset, first symbol) pairs and its stretch of the stream are staged in
//      LDS in one round of loads; every lane then finds its record by a search in LDS, cuts its window
//      out of the staged words and derives the kmer exactly like `window()` of stream_kernel.hpp.  A
//      window that covers a flagged symbol reports its element index (atomicMin): the smallest one is
//      the first element the reference would have failed on, in record order, and its first flagged
//      symbol the one it throws for.
//   3b. (round 5) tiles of reads in pool order take the DENSE path of the same kernel (below): 32-bit tile-relative indices, the
//      record of a run from a bitmap, two elements per lane and 16-byte stores side by side; from a 4-bit pool -- round 6: and from
//      DNA / RNA text -- it runs FIRST and alone, recoding its own stretches (the optimistic launch, batch_api.hip), and step 1
//      happens only if a tile asks for it.  Round 6: a symbol the kmer alphabet cannot encode (an N in a read) costs the ELEMENTS
//      whose windows hold it, not the tile: the dense path carries one flag bit per staged symbol and writes the all-ones
//      sentinel (KMERS_BATCH_SKIP) or reports the first failing element itself.
#pragma once
#include <type_traits>

#include "wide_kernel.hpp"

namespace kmers {

#ifndef KMERS_RG_RUN
#define KMERS_RG_RUN 4
#endif
constexpr int RG_RUN = KMERS_RG_RUN;     // consecutive elements per lane and pass (one-word kmers: 32 contiguous bytes per lane and array; 64 write at 60 % of that rate, profiles/r01_tuning.md)
constexpr int RG_UNIT = 1024;            // tile lengths are given in units of this many elements (KMERS_PARAM_BATCH_PASSES)
constexpr int RG_PASS = 256 * RG_RUN;    // elements per workgroup and pass of the run path; a tile is a multiple of it, 1..RG_MAX_PASSES units
#ifndef KMERS_RG_MAX_PASSES
#define KMERS_RG_MAX_PASSES 16
#endif
constexpr int RG_MAX_PASSES = KMERS_RG_MAX_PASSES;  // (the dense path's prefix scan takes up to 128 bitmap words: 16 passes.  Round 5 measured 8 and 16 the same;
                                         // with a tile's loads in one round and eight tiles per CU, round 6: 8 -> 12 -> 16 = 0.762 -> 0.773 -> 0.775 from a
                                         // 4-bit pool, 0.733 -> 0.759 -> 0.765 from text, profiles/r06_batch.md)
#ifndef KMERS_RG_DENSE_RUN
#define KMERS_RG_DENSE_RUN 2
#endif
constexpr int RG_DENSE_RUN = KMERS_RG_DENSE_RUN;  // the dense tile path: consecutive elements per lane and pass.  TWO: one 16-byte store per lane, array and pass, the
                                         // lanes' stores side by side -- every store instruction writes 1 KiB of whole cache lines.  Four (32 contiguous bytes per
                                         // lane, two half-line stores) writes at 5.4 TB/s with NO arithmetic at all (profiles/r05_batch.md)
constexpr int RG_SLOTS = 448;            // records of a tile staged in LDS (more: the global-search path)
constexpr int RG_STAGE = 1024;           // stream words of a tile staged in LDS (records in pool order: the usual case)

struct RaggedTile {              // one per tile, written by the layout pass: everything the element kernel needs to start
    uint64_t q_lo;               // ALL of its loads at once (slice of the record table, stream words, flag words).
    uint64_t f_lo;               // Stream words [q_lo, q_lo + n_words) and flag words [f_lo, f_lo + n_fwords) hold the windows
    uint32_t r_lo;               // of the tile's first and last element and what lies between them in the pool; n_words = 0
    uint32_t n_el;               // if that stretch is not ascending or longer than RG_STAGE words.  r_lo: the record that
    uint16_t n_slots;            // owns the tile's first element; n_slots: record slots to stage (up to the owner of the next
    uint16_t n_words;            // tile's first element, plus one; at most RG_SLOTS + 1).  n_el: the tile's elements -- 0 for
    uint16_t n_fwords;           // a tile past the batch's end (a grid sized for the caller's capacity), or when the batch does
    uint16_t unused;             // not fit that capacity: the element kernel never has to wait for the count itself.
};
static_assert(sizeof(RaggedTile) == 32 && RG_SLOTS + 1 < 65536 && RG_STAGE < 65536, "one 32-byte descriptor per tile");

#ifdef KMERS_RG_PROBE  // diagnostic build only (tools/ragged_probe.py): time stamps of the kernel's phases, lane 0 of every wavefront
__device__ unsigned long long rg_probe[16];
#define RG_PROBE(i)                                                                                       \
    do {                                                                                                  \
        if ((threadIdx.x & 63u) == 0 && (blockIdx.x & 127u) == 5u) {                                      \
            if ((i) == 0) rg_t0 = clock64();                                                              \
            else atomicAdd(&rg_probe[i], (unsigned long long)(clock64() - rg_t0));                        \
            if ((i) == 6) atomicAdd(&rg_probe[0], 1ull);                                                  \
        }                                                                                                 \
    } while (0)
#else
#define RG_PROBE(i) do {} while (0)
#endif

struct RaggedSpan {              // == kmers_span of the C ABI
    uint64_t first_base;
    uint64_t n_bases;
};

struct RaggedArgs {
    const uint64_t *stream;      // DST-bit little-endian symbol stream (the pool itself for Copyable pools)
    const uint64_t *flags;       // one bit per stream symbol, NULL when nothing can fail (Copyable)
    const uint64_t *any_flag;    // set by the recode pass if it flagged any symbol: clean pools skip the flag loads
    uint64_t stream_origin;      // stream symbol index of pool symbol 0
    const uint64_t *rec_off;     // [n + 1] element offset of every record (records shorter than K own nothing)
    const RaggedSpan *spans;     // [n] the records
    const RaggedTile *tiles;     // [n_tiles]
    uint64_t n_records;
    uint64_t n_elems;
    uint64_t *out_a;             // FW: forward kmers, CANON: canonical kmers
    uint64_t *out_b;             // FW: reverse complements (nullable), CANON: fx_hash (nullable)
    uint64_t seed;
    unsigned long long *err_slot;  // atomicMin of the first failing ELEMENT index
    uint32_t k;
    uint32_t skip;               // 1: elements whose window holds a flagged symbol are written as all-ones instead of failing
    uint32_t tile;               // output elements per workgroup: a multiple of RG_PASS
    uint32_t stride;             // symbols between the windows of a record: 1, or J of SpacedKmers{A,K,J} (forward kmers only)
    int32_t dense;               // KMERS_PARAM_BATCH_DENSE: -1 = never take the dense tile path (A/B, tests)
    // The OPTIMISTIC launch straight from a 4-bit pool or from DNA / RNA text (batch_api.hip): no recode pass has run, `stream` is
    // NULL, every tile tries the dense path and recodes its stretch of `src_opt` itself (one verdict per word, device_bits.hpp /
    // text8_codes; the flag of every symbol only where a word is off); a tile that is not dense writes nothing, sets its status
    // byte and counts itself.  If any did, the host runs the recode pass and launches again with `stream` set: tiles whose status
    // is 0 are done and leave at once.
    const uint64_t *src_opt;
    uint64_t n_src_words;        // the pool's words: nothing is read beyond them
    uint32_t opt_from;           // 4: a 4-bit pool; 8: text (one byte per symbol)
    uint32_t text;               // opt_from == 8: 1 = DNA text (T), 2 = RNA text (U)
    uint8_t *tile_status;
    unsigned long long *redo_count;
};

struct RecodeArgs {
    const uint64_t *src;         // pool words; word 0 holds pool symbol 0 at symbol offset (stream_origin)
    uint64_t n_words;
    uint64_t *stream;
    uint64_t *flags;
    uint64_t *any_flag;          // becomes non-zero if any symbol of the pool cannot be encoded
    uint64_t w_first;            // the launch recodes the words [w_first, w_first + n_words) (a host pool arriving in pieces, batch_api.hip)
    uint32_t ascii_table;
};

// ---- the ragged layout, computed on the device (a batch may hold tens of millions of records): scan_kernels.hpp turns the spans
// into element offsets; then one descriptor per tile.  owner(e) = the LAST record i with off[i] <= e (records that own nothing share
// their offset with the next one and are skipped by "last").
__global__ __launch_bounds__(256) void ragged_tiles_kernel(const uint64_t *off, const RaggedSpan *__restrict__ spans,
                                                            uint64_t n, uint64_t n_tiles, uint64_t n_elems_arg, const uint64_t *n_elems_ptr,
                                                            uint32_t tile, uint32_t k, uint32_t step, uint32_t dst_bits, uint64_t stream_origin,
                                                            RaggedTile *__restrict__ tiles, uint8_t *__restrict__ status,
                                                            unsigned long long *__restrict__ redo_count,
                                                            const unsigned long long *__restrict__ abort_word, uint64_t epoch, uint64_t capacity) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tiles) return;
    // (the optimistic launch's status bytes and its count of tiles left over start at zero: cleared here, on the way, instead of by
    // two fills of their own)
    if (status) status[t] = 0;
    if (redo_count && t == 0) *redo_count = 0;
    // A call that learns the count and the verdict of its layout pass only at its end (batch_api.hip: n_elems_ptr, abort_word): tiles
    // past the real count hold nothing (the grid was sized for the caller's capacity); nor does any tile if the batch does not fit
    // that capacity (KMERS_E_CAPACITY) or the look-back gave up (the offsets are not to be trusted) -- the element kernel writes nothing.
    const uint64_t n_elems = n_elems_ptr ? *n_elems_ptr : n_elems_arg;
    if (t * tile >= n_elems || (n_elems_ptr && (n_elems > capacity || *abort_word == epoch))) {
        tiles[t] = RaggedTile{};
        return;
    }
    // off[0] = 0 <= e < off[n] = n_elems.  The search starts from a HINT and gallops away from it in doubling steps before it
    // bisects: reads of one length put the hint (e's share of the records) on the owner or beside it -- two or three dependent loads
    // where the bisection of 8 M offsets took 23, three times per tile (30 us for 93 k tiles, all of it latency).
    auto owner = [&](uint64_t e, uint64_t hint) -> uint64_t {
        uint64_t lo, hi, hop = 1;  // off[lo] <= e < off[hi]
        if (off[hint] <= e) {
            lo = hint;
            hi = n;
            while (lo + hop < n) {
                if (off[lo + hop] <= e) {
                    lo += hop;
                    hop <<= 1;
                } else {
                    hi = lo + hop;
                    break;
                }
            }
        } else {
            hi = hint;
            lo = 0;
            while (hi > hop) {
                if (off[hi - hop] > e) {
                    hi -= hop;
                    hop <<= 1;
                } else {
                    lo = hi - hop;
                    break;
                }
            }
        }
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (off[mid] <= e) lo = mid;
            else hi = mid;
        }
        return lo;
    };
    auto share = [&](uint64_t e) -> uint64_t {  // the record that would own e if all records were equally long
        const uint64_t g = (uint64_t)((double)e / (double)n_elems * (double)n);
        return g < n ? g : n - 1;
    };
    const uint64_t e0 = t * tile;
    const uint64_t e_last = (e0 + tile < n_elems ? e0 + tile : n_elems) - 1;
    const uint64_t r_lo = owner(e0, share(e0));
    const uint64_t r_hi = owner(e_last, share(e_last) > r_lo ? share(e_last) : r_lo);
    const uint64_t r_next = e_last + 1 < n_elems ? owner(e_last + 1, r_hi) : n - 1;  // the last tile's slice ends with off[n]
    const uint64_t want = r_next - r_lo + 2;
    RaggedTile d{};
    d.r_lo = (uint32_t)r_lo;
    d.n_el = (uint32_t)(e_last - e0 + 1);
    d.n_slots = want < (uint64_t)(RG_SLOTS + 1) ? (uint16_t)want : (uint16_t)(RG_SLOTS + 1);
    const uint64_t p_lo = spans[r_lo].first_base + (e0 - off[r_lo]) * step + stream_origin;          // first symbol of the first window
    const uint64_t p_hi = spans[r_hi].first_base + (e_last - off[r_hi]) * step + k + stream_origin;  // one past the last window
    d.q_lo = (p_lo * dst_bits) >> 6;
    d.f_lo = p_lo >> 6;
    d.n_words = d.n_fwords = 0;
    if (p_hi > p_lo) {
        const uint64_t nw = ((p_hi * dst_bits + 63) >> 6) - d.q_lo;
        if (nw <= (uint64_t)RG_STAGE) {
            d.n_words = (uint16_t)nw;
            d.n_fwords = (uint16_t)(((p_hi + 63) >> 6) - d.f_lo);
        }
    }
    tiles[t] = d;
}

// one source word -> its stream chunk + flag bits.  Stream/flag layout per source word wi:
//   SRC 4 -> DST 2: dword wi of stream, 16 flag bits (uint16 wi)
//   SRC 8 -> DST 2: uint16 wi of stream, 8 flag bits (uint8 wi);   SRC 8 -> DST 4: dword wi, uint8 wi
//   SRC 2 -> DST 4: qwords 2wi, 2wi+1; no flags
template <int SRC, int DST>
__global__ __launch_bounds__(256) void recode_kernel(const RecodeArgs a) {
    __shared__ uint8_t lut[SRC == 8 ? 256 : 1];
    if constexpr (SRC == 8) {
        for (uint32_t i = threadIdx.x; i < 256u; i += 256u) lut[i] = ascii_entry(a.ascii_table, i);
        block_sync();
    }
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t wi = a.w_first + (uint64_t)blockIdx.x * 256u + threadIdx.x; wi < a.w_first + a.n_words; wi += stride) {
        const uint64_t x = a.src[wi];
        if constexpr (SRC == 4 && DST == 2) {
            uint32_t any_bad;  // (one verdict per word; the flag of every symbol only where a symbol is off: device_bits.hpp)
            reinterpret_cast<uint32_t *>(a.stream)[wi] = pack_4to2_checked(x, any_bad);
            reinterpret_cast<uint16_t *>(a.flags)[wi] = any_bad ? (uint16_t)bad_bits16(bad_nibbles4(x)) : (uint16_t)0;
            if (any_bad) *a.any_flag = 1;  // rare; any writer, same value
        } else if constexpr (SRC == 8) {
            uint32_t codes = 0, f = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t v = lut[(x >> (8 * j)) & 0xffu];
                codes |= (v & 0xfu) << (DST * j);
                f |= (v >> 7) << j;
            }
            if constexpr (DST == 2) reinterpret_cast<uint16_t *>(a.stream)[wi] = (uint16_t)codes;
            else reinterpret_cast<uint32_t *>(a.stream)[wi] = codes;
            reinterpret_cast<uint8_t *>(a.flags)[wi] = (uint8_t)f;
            if (f) *a.any_flag = 1;
        } else {  // SRC 2 -> DST 4 (TwoToFour): 1 << code, nothing can fail
            a.stream[2 * wi] = expand_2to4((uint32_t)x);
            a.stream[2 * wi + 1] = expand_2to4((uint32_t)(x >> 32));
        }
    }
}

// is any of the k flag bits that start at bit fs of flag word load(0) set?  (k up to 128: up to three words)
template <class Load>
__device__ __forceinline__ bool any_flag_in(Load load, uint32_t fs, uint32_t k) {
    uint64_t f = load(0u) >> fs;
    const uint32_t have = 64u - fs;
    if (have >= k) return (k < 64u ? (f & ((1ull << k) - 1ull)) : f) != 0;
    uint32_t left = k - have;
    for (uint32_t i = 1; left; ++i) {
        uint64_t w = load(i);
        if (left < 64u) {
            w &= (1ull << left) - 1ull;
            left = 0;
        } else {
            left -= 64u;
        }
        f |= w;
    }
    return f != 0;
}

// `window()` of stream_kernel.hpp with the stream words coming from `load(j)` = word j of the stream counted
// from the one that holds `bit` (LDS or HBM)
template <int N, int DST, class Load>
__device__ __forceinline__ void window_words(Load load, uint32_t s, uint32_t k, uint64_t mask, uint64_t (&fw)[N], uint64_t (&rc)[N]) {
    uint64_t W[N], R[N];
    uint64_t lo = load(0u);
    const uint32_t need = (s + (uint32_t)DST * k + 63u) >> 6;  // stream words the window touches (<= N + 1)
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const uint64_t hi = (uint32_t)(j + 1) < need ? load((uint32_t)(j + 1)) : 0;  // never read past the window's last word
        W[j] = funnel64(lo, hi, s);
        lo = hi;
    }
    W[N - 1] &= mask;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        rc[N - 1 - j] = comp_symbols<DST>(W[j]);
        R[j] = rev_symbols<DST>(W[j]);
    }
    if constexpr (DST == 2) rc[0] &= mask;
    const uint32_t sh = 64u * N - (uint32_t)DST * k;
    fw[0] = R[0] >> sh;
#pragma unroll
    for (int i = 1; i < N; ++i) fw[i] = (R[i] >> sh) | ((R[i - 1] << 1) << (63u - sh));
}

// ---- the DENSE tile path (round 5; profiles/r05_batch.md) -----------------------------------------------------------------
// ragged_kernel's general element loop costs 84 vector instructions per element on the bench's batch (8 M reads x 125 bases,
// SQ_INSTS_VALU = 1.0e9 wave instructions for 760 M elements) at 5.2 cycles of a SIMD each: the launch is bound by instruction
// issue (2.6 ms), not by its 12.2 GB of stores (1.75 ms at the two-class write rate).  What costs: 64-bit element and symbol
// indices everywhere, a binary search per run, a second window cut in every wavefront (one lane in 24 crosses a record boundary,
// so 87 % of the wavefronts pay for both cuts), the flag logic of pools that hold no flagged symbol, per-access "is it staged"
// tests.  A tile is DENSE when none of that is needed: every record of it owns at least RG_DENSE_RUN elements (a run crosses at most one
// boundary and no record is empty), its records lie inside the staged stretch of the stream, and no symbol of the pool is flagged.
// Then, with every index relative to the tile and 32 bits wide:
//   * the record of a run comes from a BITMAP of the run indices at which a record begins (one bit per run, set with LDS atomics
//     from the staged offsets) and a prefix count per 64 runs: one broadcast read and a population count per run, no search;
//   * every lane cuts its run's first window and rolls the rest (CanonicalKmers.jl:131-144); only a lane whose run crosses into
//     the next record cuts that record's first window as well;
//   * every run leaves with two 16-byte stores per array: the launch waits for its store instructions, not for their bytes.
// Same results as the general path: tests/test_gpu_batch.py and the batch fuzz run both (KMERS_PARAM_BATCH_DENSE = -1 forces the
// general path).
template <int DST>
__device__ __forceinline__ void dense_cut(const uint64_t *src_l, uint32_t p, uint32_t k, uint64_t mask, uint64_t &fw, uint64_t &rc, uint32_t &next_syms) {
    const uint32_t bit = p * (uint32_t)DST, q = bit >> 6, sh = bit & 63u;
    const uint64_t l0 = src_l[q], l1 = src_l[q + 1u], l2 = src_l[q + 2u];  // (the staged array is two words longer than the stretch)
    const uint64_t W0 = funnel64(l0, l1, sh), W1 = funnel64(l1, l2, sh);
    const uint64_t W = W0 & mask;
    fw = rev_symbols<DST>(W) >> (64u - (uint32_t)DST * k);
    rc = comp_symbols<DST>(W);
    if constexpr (DST == 2) rc &= mask;
    next_syms = (uint32_t)((uint32_t)DST * k == 64u ? W1 : funnel64(W0, W1, (uint32_t)DST * k));  // symbols K, K + 1, ... of the sub-run
}
template <int DST>
__device__ __forceinline__ void dense_roll(uint64_t &fw, uint64_t &rc, uint32_t syms, uint32_t j, uint64_t mask, uint32_t top) {
    const uint64_t sym = (syms >> ((uint32_t)DST * j)) & ((1u << DST) - 1u);  // the (j + 1)-th symbol after the first window
    uint64_t csym;
    if constexpr (DST == 2) csym = sym ^ 3u;
    else csym = ((sym & 1u) << 3) | ((sym & 2u) << 1) | ((sym & 4u) >> 1) | ((sym & 8u) >> 3);
    fw = ((fw << DST) | sym) & mask;       // shift_encoding, construction_utils.jl:129-134
    rc = (rc >> DST) | (csym << top);      // shift_first_encoding of the complement, kmer.jl:511-518
}
template <int MODE>
__device__ __forceinline__ void dense_finish(uint64_t fw, uint64_t rc, uint64_t seed_rot, uint64_t &x, uint64_t &y) {
    if constexpr (MODE == MODE_FW) {
        x = fw;
        y = rc;
    } else {
        x = fw < rc ? fw : rc;             // CanonicalKmers.jl:220-225
        y = (seed_rot ^ x) * FX_CONSTANT;  // fx_hash of one word, kmer.jl:255-261 (seed_rot = rotl(seed, 5))
    }
}

struct DenseLds {                        // carved out of ragged_kernel's LDS arrays
    uint64_t *slot;                      // [RG_SLOTS + 1]: low half = the record's first element relative to the tile (clamped to [0, 2^31)),
                                         //   high half = first stream symbol of element 0 of the TILE if it belonged to this record, relative to the staged stretch
    uint64_t *src;                       // [RG_STAGE + 2]
    uint64_t *flg;                       // [RG_STAGE / 2 + 4]: one "cannot be encoded" bit per symbol of the stretch (round 6)
    uint64_t *bits;                      // [RG_MAX_PASSES * RG_UNIT / RG_RUN / 64]: bit j: a record begins in run j or between run j - 1's first element and it
    uint32_t *base;                      // [... + 1]: records that begin before the word's first run
};

// the eight "byte j is off" bits of text8_bad_bytes (bit 8j) gathered into one byte (the partial products never meet: no carries)
__device__ __forceinline__ uint32_t gather_byte_flags(uint64_t f) { return (uint32_t)((f * 0x0102040810204080ull) >> 56); }

// The runs of a dense tile.  FLAGGED: the tile's stretch holds a symbol the kmer alphabet cannot encode -- every run looks its
// windows' flag bits up (bit fbase + p of L.flg = symbol p of the stretch) and either writes the all-ones sentinel for the
// elements that hold one (a.skip: KMERS_BATCH_SKIP, UnambiguousKmers' selection at the strict call's indices,
// src/iterators/UnambiguousKmers.jl:109-148) or reports the first of them (the reference throws there, FwKmers.jl:112).
template <int DST, int MODE, bool FLAGGED>
__device__ __forceinline__ void dense_runs(const RaggedArgs &a, const DenseLds &L, uint64_t e0, uint32_t n_el, uint32_t fbase) {
    constexpr uint32_t RUN = RG_DENSE_RUN;
    const uint32_t tid = threadIdx.x, k = a.k;
    const uint64_t mask = head_mask((int)k, DST);
    const uint32_t top = (uint32_t)DST * (k - 1u);
    const uint64_t seed_rot = (a.seed << 5) | (a.seed >> 59);
    const uint64_t kbits = k >= 64u ? ~0ull : (1ull << k) - 1ull;
    auto flag_bits = [&](uint32_t p, uint32_t span) -> uint64_t {  // flags of the symbols [p, p + span) of the stretch (span <= 33)
        const uint32_t b = p + fbase;
        return funnel64(L.flg[b >> 6], L.flg[(b >> 6) + 1u], b & 63u) & ((1ull << span) - 1ull);
    };
    for (uint32_t j = tid; j * RUN < n_el; j += 256u) {
        const uint32_t e = j * RUN;
        const uint64_t word = L.bits[j >> 6];               // (the same word for the whole wavefront)
        const uint32_t lane = j & 63u;
        const uint32_t r = L.base[j >> 6] + (uint32_t)__popcll(word & ((2ull << lane) - 1ull));  // records that begin at or before element e
        const uint64_t s0 = L.slot[r], s1 = L.slot[r + 1u];
        const uint32_t left_in_tile = n_el - e;
        const uint32_t to_boundary = (uint32_t)s1 - e;      // > 0: slot r + 1 begins behind e
        const uint32_t cnt = left_in_tile < RUN ? left_in_tile : RUN;
        const uint32_t lenA = to_boundary < cnt ? to_boundary : cnt;
        uint64_t fw, rc, X[RUN], Y[RUN];
        uint32_t syms;
#if defined(KMERS_RG_CUT) && KMERS_RG_CUT == 2  // diagnostic build: the stores and the lookups, no window arithmetic
        for (uint32_t t = 0; t < RUN; ++t) {
            X[t] = s0 + t;
            Y[t] = s1 + lenA;
        }
        if (true) {
            const uint64_t g2 = e0 + e;
            for (uint32_t t = 0; t < RUN; t += 2) *reinterpret_cast<ulonglong2 *>(a.out_a + g2 + t) = make_ulonglong2(X[t], X[t + 1]);
            for (uint32_t t = 0; t < RUN; t += 2) *reinterpret_cast<ulonglong2 *>(a.out_b + g2 + t) = make_ulonglong2(Y[t], Y[t + 1]);
            continue;
        }
#endif
        const uint32_t pA = (uint32_t)(s0 >> 32) + e, pB = (uint32_t)(s1 >> 32) + e + lenA;
        const uint64_t g = e0 + e;
        dense_cut<DST>(L.src, pA, k, mask, fw, rc, syms);
        // a lane whose run crosses into the next record cuts that record's first window too (one lane in 24 at 95 kmers per read:
        // most wavefronts run this once, for a few lanes) -- every lane then holds a whole run and stores it with 16-byte stores
        // (a version that left the sub-runs to a list and stored them element by element had half the vector instructions of
        // this one and was SLOWER: 2.4 times the store instructions, and those are what the launch waits for)
        uint64_t fwB = 0, rcB = 0;
        uint32_t symsB = 0;
        if (lenA < cnt) dense_cut<DST>(L.src, pB, k, mask, fwB, rcB, symsB);
        uint64_t fA = 0, fB = 0;                            // flagged symbols of the two sub-runs
        if constexpr (FLAGGED) {
            fA = flag_bits(pA, k + lenA - 1u);
            if (lenA < cnt) fB = flag_bits(pB, k + (cnt - lenA) - 1u);
            if (!a.skip) {                                  // the first element whose window holds a flagged symbol (outputs are unspecified then)
                if (fA) {
                    const uint32_t first = (uint32_t)__builtin_ctzll(fA);
                    atomicMin(a.err_slot, (unsigned long long)(g + (first >= k ? first - k + 1u : 0u)));
                } else if (fB) {
                    const uint32_t first = (uint32_t)__builtin_ctzll(fB);
                    atomicMin(a.err_slot, (unsigned long long)(g + lenA + (first >= k ? first - k + 1u : 0u)));
                }
            }
        }
        dense_finish<MODE>(fw, rc, seed_rot, X[0], Y[0]);
#pragma unroll
        for (uint32_t t = 1; t < RUN; ++t) {
            if (t == lenA) {                                // the next record's first element
                fw = fwB;
                rc = rcB;
                syms = symsB << ((uint32_t)DST * t);        // (its following symbols are indexed from t on below)
            } else {
                dense_roll<DST>(fw, rc, syms, t - 1u, mask, top);
            }
            dense_finish<MODE>(fw, rc, seed_rot, X[t], Y[t]);
        }
        if constexpr (FLAGGED) {
            if (a.skip) {
#pragma unroll
                for (uint32_t t = 0; t < RUN; ++t) {
                    const uint64_t in_window = t < lenA ? (fA >> t) & kbits : (fB >> (t - lenA)) & kbits;
                    if (in_window) X[t] = Y[t] = ~0ull;     // never a canonical kmer, never a (kmer, reverse complement) pair
                }
            }
        }
#if defined(KMERS_RG_CUT) && KMERS_RG_CUT == 1  // diagnostic build: everything but the stores (profiles/r05_batch.md)
        {
            uint64_t all = 0;
            for (uint32_t t = 0; t < RUN; ++t) all ^= X[t] ^ Y[t];
            if (all != 0x0123456789abcdefull) continue;
        }
#endif
        if (cnt == RUN) {
            if (a.out_a) {
#pragma unroll
                for (uint32_t t = 0; t < RUN; t += 2) *reinterpret_cast<ulonglong2 *>(a.out_a + g + t) = make_ulonglong2(X[t], X[t + 1]);
            }
            if (a.out_b) {
#pragma unroll
                for (uint32_t t = 0; t < RUN; t += 2) *reinterpret_cast<ulonglong2 *>(a.out_b + g + t) = make_ulonglong2(Y[t], Y[t + 1]);
            }
        } else {                                            // the end of the batch inside the run
#pragma unroll
            for (uint32_t t = 0; t < RUN; ++t) {
                if (t < cnt) {
                    if (a.out_a) a.out_a[g + t] = X[t];
                    if (a.out_b) a.out_b[g + t] = Y[t];
                }
            }
        }
    }
}

// true: the tile was written.  false (uniformly): the tile is not dense, nothing was written, the general path must run.
// FROM: 0 = the recoded stream (and `flags`, the recode pass's flag words, or NULL); 4 / 8 = the optimistic launch, the tile
// recodes its own stretch of a 4-bit pool / of DNA or RNA text and makes its own flag bits.
template <int DST, int MODE, int FROM>
__device__ __forceinline__ bool ragged_dense_tile(const RaggedArgs &a, const RaggedTile &d, uint64_t e0, uint32_t n_el, const DenseLds &L,
                                                  const uint64_t *flags) {
    constexpr uint32_t RUN = RG_DENSE_RUN, SPW = 64u / (uint32_t)DST;
    static_assert(FROM == 0 || DST == 2, "the optimistic launch yields 2-bit kmers");
    const uint32_t tid = threadIdx.x, k = a.k, n_rec = d.n_slots;
    const uint64_t sym0 = d.q_lo * SPW;                     // stream symbol at bit 0 of the staged stretch
    const uint32_t staged_syms = d.n_words * SPW;
    // ---- stage the record slots, the stream stretch (and its flag bits); test the tile
    uint32_t bad = (d.n_words == 0u || n_rec > (uint32_t)RG_SLOTS + 1u || n_rec < 2u) ? 1u : 0u;
    uint32_t flagged = 0, n_flg = 0, fbase = 0;
    // (the stretch of the stream first: its loads are in flight beside those of the record slots -- one round trip, not two)
    // the record slots: two steps per lane cover the RG_SLOTS + 1 slots; their loads are issued together with the stretch's
    uint64_t slot_o[2] = {0, 0}, slot_next[2] = {0, 0};
    RaggedSpan slot_sp[2] = {{0, 0}, {0, 0}};
    auto slots_load = [&]() {
#pragma unroll
        for (uint32_t t = 0; t < 2u; ++t) {
            const uint32_t i = tid + t * 256u;
            if (i < n_rec) {
                const uint64_t rec = d.r_lo + (uint64_t)i;
                const bool real = rec < a.n_records;
                slot_o[t] = a.rec_off[rec];
                slot_next[t] = (i + 1u < n_rec && real) ? a.rec_off[rec + 1] : 0;
                slot_sp[t] = real ? a.spans[rec] : RaggedSpan{0, 0};
            }
        }
    };
    auto slot_stage = [&](uint32_t i, uint64_t o, uint64_t o_next, const RaggedSpan &sp) {
        const bool real = d.r_lo + (uint64_t)i < a.n_records;
        const uint32_t rel = o <= e0 ? 0u : (o - e0 > 0x7fffffffull ? 0x7fffffffu : (uint32_t)(o - e0));
        const uint32_t delta = (uint32_t)(sp.first_base + a.stream_origin + e0 - o - sym0);
        L.slot[i] = (uint64_t)rel | ((uint64_t)delta << 32);
        if (i + 1u == n_rec && rel < n_el) bad = 1u;        // more records than slots: the staged slice does not close the tile
        if (i + 1u < n_rec && real) {                       // (the last slot only closes the search range)
            const uint64_t lo_e = o > e0 ? o : e0, hi_e = o_next < e0 + n_el ? o_next : e0 + n_el;  // its elements inside the tile
            if (o_next - o < (uint64_t)RUN) bad = 1u;       // a record that owns fewer than RUN elements (or none)
            if (hi_e > lo_e) {                              // its windows must lie in the staged stretch (RUN - 1 symbols of slack for the roll)
                const uint64_t first = sp.first_base + a.stream_origin + (lo_e - o), end = sp.first_base + a.stream_origin + (hi_e - 1u - o) + k;
                if (first < sym0 || end > sym0 + staged_syms) bad = 1u;
            }
        }
    };
    // A tile's lifetime is what the launch is made of (about seven tiles per CU at a time): every round trip to HBM in front of the
    // stores counts, and under the launch's own store traffic a round takes microseconds.  So ALL loads of the stretch are issued
    // before any of them is used -- up to STEPS steps of two source words per lane held in registers (a longer stretch takes the
    // plain loop for the rest) -- and 16 bytes at a time where the pool's base allows it.
    if constexpr (FROM == 4 || FROM == 8) {
        constexpr uint32_t PER = FROM == 4 ? 2u : 4u;     // source words per stream word
        constexpr uint64_t FILL = FROM == 4 ? 0x1111111111111111ull : 0x4141414141414141ull;  // "A": in no window, never flagged
        constexpr uint32_t STEPS = FROM == 4 ? 3u : 6u;   // 768 / 1536 pairs of words: stretches of 768 stream words (24 k symbols: 16 k elements of 125-base reads)
        const uint64_t w0 = (uint64_t)PER * d.q_lo;       // (a stretch may end past the pool's last word: FILL there)
        const uint64_t want = (uint64_t)PER * d.n_words;
        const uint32_t avail = w0 < a.n_src_words ? (uint32_t)(a.n_src_words - w0 < want ? a.n_src_words - w0 : want) : 0u;
        const uint64_t *src = a.src_opt + w0;
        const uint32_t n_pairs = (uint32_t)want / 2u;
        const uint32_t n_pad = FROM == 4 ? (n_pairs + 1u) & ~1u : (n_pairs + 3u) & ~3u;  // (whole flag words: the padding is unflagged)
        const bool wide = (reinterpret_cast<uintptr_t>(src) & 15u) == 0;
        auto load_pair = [&](uint32_t i, uint64_t &x0, uint64_t &x1) {
            const uint32_t w = 2u * i;
            if (wide && w + 1u < avail) {
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(src + w);
                x0 = v.x;
                x1 = v.y;
            } else {
                x0 = w < avail ? src[w] : FILL;
                x1 = w + 1u < avail ? src[w + 1u] : FILL;
            }
        };
        auto stage_pair = [&](uint32_t i, uint64_t x0, uint64_t x1) {
            if constexpr (FROM == 4) {
                uint32_t bad0, bad1;
                const uint32_t c0 = pack_4to2_checked(x0, bad0), c1 = pack_4to2_checked(x1, bad1);  // FourToTwo, construction_utils.jl:47-52
                if (i < n_pairs) L.src[i] = (uint64_t)c0 | ((uint64_t)c1 << 32);
                uint32_t f = 0;                     // a symbol that is not one-hot: which ones
                if (bad0 | bad1) f = (bad0 ? bad_bits16(bad_nibbles4(x0)) : 0u) | ((bad1 ? bad_bits16(bad_nibbles4(x1)) : 0u) << 16);
                reinterpret_cast<uint32_t *>(L.flg)[i] = f;
                flagged |= f;
            } else {
                uint32_t off0, off1;
                const uint32_t c0 = text8_codes(x0, a.text, off0), c1 = text8_codes(x1, a.text, off1);  // AsciiEncode, src/construction.jl:94-95
                if (i < n_pairs) reinterpret_cast<uint32_t *>(L.src)[i] = c0 | (c1 << 16);
                uint32_t f = 0;
                if (off0 | off1) f = (off0 ? gather_byte_flags(text8_bad_bytes(x0, a.text)) : 0u) | ((off1 ? gather_byte_flags(text8_bad_bytes(x1, a.text)) : 0u) << 8);
                reinterpret_cast<uint16_t *>(L.flg)[i] = (uint16_t)f;
                flagged |= f;
            }
        };
        uint64_t x[STEPS][2];
#pragma unroll
        for (uint32_t t = 0; t < STEPS; ++t) {
            x[t][0] = x[t][1] = FILL;
            if (tid + t * 256u < n_pad) load_pair(tid + t * 256u, x[t][0], x[t][1]);
        }
        slots_load();                                       // (in flight beside the stretch's words: one round trip, not two)
#pragma unroll
        for (uint32_t t = 0; t < STEPS; ++t)
            if (tid + t * 256u < n_pad) stage_pair(tid + t * 256u, x[t][0], x[t][1]);
        for (uint32_t i = tid + STEPS * 256u; i < n_pad; i += 256u) {
            uint64_t x0, x1;
            load_pair(i, x0, x1);
            stage_pair(i, x0, x1);
        }
        n_flg = FROM == 4 ? n_pad / 2u : n_pad / 4u;
    } else {
        uint64_t y[2] = {0, 0}, g[1] = {0};
#pragma unroll
        for (uint32_t t = 0; t < 2u; ++t)
            if (tid + t * 256u < d.n_words) y[t] = a.stream[d.q_lo + tid + t * 256u];
        if (flags && tid < d.n_fwords) g[0] = flags[d.f_lo + tid];
        slots_load();
#pragma unroll
        for (uint32_t t = 0; t < 2u; ++t)
            if (tid + t * 256u < d.n_words) L.src[tid + t * 256u] = y[t];
        for (uint32_t i = tid + 512u; i < d.n_words; i += 256u) L.src[i] = a.stream[d.q_lo + i];
        if (flags) {
            if (tid < d.n_fwords) {
                L.flg[tid] = g[0];
                flagged |= g[0] != 0 ? 1u : 0u;
            }
            for (uint32_t i = tid + 256u; i < d.n_fwords; i += 256u) {
                const uint64_t w = flags[d.f_lo + i];
                L.flg[i] = w;
                flagged |= w != 0 ? 1u : 0u;
            }
            n_flg = d.n_fwords;
            fbase = (uint32_t)(sym0 - d.f_lo * 64u);        // (the flag words begin at or before the stretch)
        }
    }
    if (tid < 2u) {
        L.src[d.n_words + tid] = 0;
        L.flg[n_flg + tid] = 0;
    }
    {
#pragma unroll
        for (uint32_t t = 0; t < 2u; ++t)
            if (tid + t * 256u < n_rec) slot_stage(tid + t * 256u, slot_o[t], slot_next[t], slot_sp[t]);
    }
    constexpr uint32_t N_WORDS = RG_MAX_PASSES * RG_UNIT / RUN / 64;
    static_assert(N_WORDS <= 128 && N_WORDS % 2 == 0, "one wavefront scans the bitmap's words, two per lane");
    if (tid < N_WORDS) L.bits[tid] = 0;
    lds_atomics_settle();
    if (__syncthreads_or((int)bad)) return false;           // (also orders the staging before what follows)
    const bool any_flag = (FROM != 0 || flags) && __syncthreads_or((int)flagged) != 0;  // (uniform: both sides of the && are)
    // ---- the bitmap of record beginnings
    uint32_t *bits32 = reinterpret_cast<uint32_t *>(L.bits);
    for (uint32_t i = 1u + tid; i < n_rec; i += 256u) {
        const uint32_t rel = (uint32_t)L.slot[i];
        const uint32_t j = (rel + RUN - 1u) / RUN;          // the first run that starts at or behind the record's first element
        if (rel > 0u && j * RUN < n_el) atomicOr(&bits32[j >> 5], 1u << (j & 31u));  // (a record that begins inside the LAST run needs no bit)
    }
    block_sync();
    if (tid < 64u) {                                        // exclusive prefix of the words' population counts (one wavefront, two words per lane)
        const uint32_t c0 = 2u * tid < N_WORDS ? (uint32_t)__popcll(L.bits[2u * tid]) : 0u;
        const uint32_t c1 = 2u * tid + 1u < N_WORDS ? (uint32_t)__popcll(L.bits[2u * tid + 1u]) : 0u;
        uint32_t incl = c0 + c1;
#pragma unroll
        for (uint32_t step = 1; step < 64u; step <<= 1) {
            const uint32_t up = __shfl_up(incl, step, 64);
            if (tid >= step) incl += up;
        }
        if (2u * tid < N_WORDS) {
            L.base[2u * tid] = incl - c0 - c1;
            L.base[2u * tid + 1u] = incl - c1;
        }
    }
    block_sync();
    // ---- the runs
    if (any_flag) dense_runs<DST, MODE, true>(a, L, e0, n_el, fbase);
    else dense_runs<DST, MODE, false>(a, L, e0, n_el, fbase);
    return true;
}

// VEC: out_a / out_b are 16-byte aligned (one-word kmers: RG_RUN elements per lane and pass, 16-byte stores)
// (amdgpu_waves_per_eu(8): the one-word kernel needs 106 scalar registers as the compiler likes it, which is SEVEN wavefronts per SIMD on
// gfx9's 800-entry scalar file; asked for eight it keeps a few in vector lanes instead.  With the LDS at 19.3 KiB that is eight tiles
// per CU in flight, not seven.)
template <int DST, int N, int MODE, bool VEC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8))) void ragged_kernel(const RaggedArgs a) {
    __shared__ uint64_t off_l[RG_SLOTS + 1];    // element offset of every record slot of the tile
    __shared__ uint64_t delta_l[RG_SLOTS + 1];  // first stream symbol of the slot's record minus offset * stride: window of element g = delta + g * stride
    __shared__ uint64_t src_l[RG_STAGE + 2];
    __shared__ uint64_t flg_l[RG_STAGE / 2 + 4];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, DST);
    // (the pool may hold flagged symbols outside every record, so a set any_flag only means "look")
    const uint64_t *flags = (a.flags && *a.any_flag) ? a.flags : nullptr;
    const uint64_t tile = blockIdx.x;
    const uint64_t e0 = tile * a.tile;
#ifdef KMERS_RG_PROBE
    long long rg_t0 = 0;
#endif
    RG_PROBE(0);
    // (the descriptor says how many elements the tile holds: none past the batch's end or when the batch does not fit the caller's
    // capacity -- the kernel itself never reads the element count, one round of dependent loads less per tile)
    const RaggedTile d = a.tiles[tile];
    if (d.n_el == 0u) return;
    const uint64_t r_lo = d.r_lo;
    const uint64_t e_last = e0 + d.n_el - 1u;
    if constexpr (N == 1 && VEC) {
        // dense tiles (comment above ragged_dense_tile): one-word kmers, aligned outputs, consecutive windows
        DenseLds L;
        L.slot = off_l;
        L.src = src_l;
        L.flg = flg_l;
        // the dense path's bitmap and its prefix counts live in delta_l, which only the general path uses: 19.3 KiB of LDS per
        // workgroup = EIGHT workgroups per CU instead of seven (a tile's lifetime over the tiles in flight is what the launch takes)
        L.bits = delta_l;
        L.base = reinterpret_cast<uint32_t *>(delta_l + RG_MAX_PASSES * RG_UNIT / RG_DENSE_RUN / 64);
        static_assert(sizeof(delta_l) >= (RG_MAX_PASSES * RG_UNIT / RG_DENSE_RUN / 64) * 12 + 16, "LDS carve");
        static_assert(sizeof(off_l) + sizeof(delta_l) + sizeof(src_l) + sizeof(flg_l) <= 160 * 1024 / 8, "eight workgroups per CU");
        if constexpr (DST == 2) {
            if (a.src_opt) {  // the optimistic launch (RaggedArgs): dense or nothing
                bool done = false;
                if (a.tile <= (uint32_t)(RG_MAX_PASSES * RG_UNIT)) {
                    if (a.opt_from == 8u) done = ragged_dense_tile<DST, MODE, 8>(a, d, e0, (uint32_t)(e_last - e0 + 1), L, nullptr);
                    else done = ragged_dense_tile<DST, MODE, 4>(a, d, e0, (uint32_t)(e_last - e0 + 1), L, nullptr);
                }
                if (!done && tid == 0) {
                    a.tile_status[tile] = 1;
                    atomicAdd(a.redo_count, 1ull);
                }
                return;
            }
        }
        if (a.tile_status && !a.tile_status[tile]) return;  // (the launch after an optimistic one: this tile is done)
        if (a.stride == 1u && a.dense >= 0 && a.tile <= (uint32_t)(RG_MAX_PASSES * RG_UNIT)) {
            if (ragged_dense_tile<DST, MODE, 0>(a, d, e0, (uint32_t)(e_last - e0 + 1), L, flags)) return;
            block_sync();  // (not dense: the general path below re-stages the tile)
        }
    }
    // Everything the tile reads is requested here, in ONE round of loads (a load takes about 5 us while the
    // device is saturated with the stores of the other workgroups: rounds are what a tile's time is made of).
    // Offsets and first symbols of the tile's records: r_lo .. the record that owns the next tile's first
    // element, plus the one after it (its offset lies past this tile: the end of the search range;
    // off[n] = n_elems closes the last tile).  At most RG_SLOTS + 1 slots: a tile crowded with short records
    // is not covered, and its lanes search the global arrays instead (correct, just slower).  The stream
    // (and flag) words between the tile's first and last window: records that lie in pool order (reads,
    // contigs of one file) find all their windows there; a lane whose window lies elsewhere reads HBM.
    const uint32_t n_rec = d.n_slots;
    for (uint32_t i = tid; i < n_rec; i += 256u) {
        const uint64_t o = a.rec_off[r_lo + i];
        const uint64_t b = r_lo + i < a.n_records ? a.spans[r_lo + i].first_base : 0;
        off_l[i] = o;
        delta_l[i] = b - o * a.stride + a.stream_origin;
    }
    for (uint32_t i = tid; i < d.n_words; i += 256u) src_l[i] = a.stream[d.q_lo + i];
    if (flags)
        for (uint32_t i = tid; i < d.n_fwords; i += 256u) flg_l[i] = flags[d.f_lo + i];
    RG_PROBE(1);
    block_sync();
    RG_PROBE(2);
    const bool covered = off_l[n_rec - 1] > e_last;
    // Two separate instantiations of the element loop -- LDS lookups or global lookups -- rather than a
    // per-access select: hipcc turned `covered ? lds[i] : global[i]` into FLAT loads whose address is
    // selected between the LDS aperture and HBM, and that version produced wrong results for whole
    // wavefronts, non-deterministically, on gfx950 (tools/check_isa.sh, r01_tuning.md).
    auto run = [&](auto covered_tag) {
        constexpr bool COVERED = decltype(covered_tag)::value;
        // record slot of element g: the last slot with off <= g.  `from` = a slot known to qualify (slot 0
        // always does; a lane's elements ascend, so its previous answer does).  The last staged slot's offset
        // lies past the tile, so the search never leaves the slice.
        auto slot_of = [&](uint64_t g, uint32_t from) -> uint32_t {
            if constexpr (COVERED) {
                uint32_t lo = from, hi = n_rec - 1u;
                if (off_l[lo + 1u] > g) return lo;                 // long records: still the same one
                while (lo + 8u < hi && off_l[lo + 8u] <= g) lo += 8u;  // a pass ahead is a few records ahead
                if (lo + 8u < hi) hi = lo + 8u;
                while (hi - lo > 1u) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (off_l[mid] <= g) lo = mid;
                    else hi = mid;
                }
                return lo;
            } else {
                uint64_t lo = r_lo + from, hi = a.n_records;
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (a.rec_off[mid] <= g) lo = mid;
                    else hi = mid;
                }
                return (uint32_t)(lo - r_lo);
            }
        };
        // element g of record slot r: forward kmer and reverse complement (and the flag test)
        auto element = [&](uint64_t g, uint32_t r, uint64_t (&fw)[N], uint64_t (&rc)[N]) -> bool {
            uint64_t p;  // stream symbol index of the window
            if constexpr (COVERED) p = delta_l[r] + g * a.stride;
            else p = a.spans[r_lo + r].first_base + (g - a.rec_off[r_lo + r]) * a.stride + a.stream_origin;
            const uint64_t bit = p * (uint64_t)DST;
            const uint64_t q = bit >> 6;
            const uint32_t s = (uint32_t)(bit & 63u);
            const uint32_t need = (s + (uint32_t)DST * k + 63u) >> 6;
            const uint64_t rel = q - d.q_lo;  // (wraps for a window before the staged stretch)
            const bool staged = COVERED && rel < (uint64_t)d.n_words && rel + need <= (uint64_t)d.n_words;
            bool bad = false;
            if (flags) {
                const uint64_t fq = p >> 6;
                if (staged) {  // (the staged flag words cover the staged stream words)
                    const uint32_t fr = (uint32_t)(fq - d.f_lo);
                    bad = any_flag_in([&](uint32_t i) { return flg_l[fr + i]; }, (uint32_t)(p & 63u), k);
                } else {
                    bad = any_flag_in([&](uint32_t i) { return flags[fq + i]; }, (uint32_t)(p & 63u), k);
                }
                if (bad && !a.skip) atomicMin(a.err_slot, (unsigned long long)g);
            }
            if (staged) window_words<N, DST>([&](uint32_t j) { return src_l[(uint32_t)rel + j]; }, s, k, mask, fw, rc);
            else window_words<N, DST>([&](uint32_t j) { return a.stream[q + j]; }, s, k, mask, fw, rc);
            return bad && a.skip;  // true: the caller writes the all-ones sentinel
        };
        auto finish = [&](const uint64_t (&fw)[N], const uint64_t (&rc)[N], uint64_t (&x)[N], uint64_t (&y)[N]) {
            // x -> out_a, y -> out_b (FW: reverse complement; CANON: y[0] = fx_hash)
            if constexpr (MODE == MODE_FW) {
    #pragma unroll
                for (int w = 0; w < N; ++w) { x[w] = fw[w]; y[w] = rc[w]; }
            } else {
                const bool lt = kmer_less<N>(fw, rc);  // fw < rv ? fw : rv, CanonicalKmers.jl:220-225
    #pragma unroll
                for (int w = 0; w < N; ++w) { x[w] = lt ? fw[w] : rc[w]; y[w] = 0; }
                y[0] = fx_hash<N>(x, a.seed);
            }
        };
        auto one_by_one = [&]() {  // every lane one element at a time: multi-word kmers, unaligned outputs, uncovered tiles, strides
            uint32_t r = 0;
            if constexpr (N == 1 && VEC) {  // (one-word kmers: two neighbours per lane, one 16-byte store per array)
                for (uint32_t e = 2u * tid; e < a.tile; e += 512u) {
                    const uint64_t g = e0 + e;
                    if (g > e_last) break;
                    uint64_t fw[1], rc[1], x0[1], y0[1], x1[1] = {0}, y1[1] = {0};
                    r = slot_of(g, r);
                    const bool m0 = element(g, r, fw, rc);
                    finish(fw, rc, x0, y0);
                    if (m0) x0[0] = y0[0] = ~0ull;
                    const bool two = g + 1 <= e_last;
                    if (two) {
                        const bool m1 = element(g + 1, slot_of(g + 1, r), fw, rc);
                        finish(fw, rc, x1, y1);
                        if (m1) x1[0] = y1[0] = ~0ull;
                        if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(x0[0], x1[0]);
                        if (a.out_b) *reinterpret_cast<ulonglong2 *>(a.out_b + g) = make_ulonglong2(y0[0], y1[0]);
                    } else {
                        if (a.out_a) a.out_a[g] = x0[0];
                        if (a.out_b) a.out_b[g] = y0[0];
                    }
                }
                return;
            }
            for (uint32_t e = tid; e < a.tile; e += 256u) {
                const uint64_t g = e0 + e;
                if (g > e_last) break;
                uint64_t fw[N], rc[N], x[N], y[N];
                r = slot_of(g, r);
                const bool masked = element(g, r, fw, rc);
                finish(fw, rc, x, y);
                if (masked) {
#pragma unroll
                    for (int w = 0; w < N; ++w) x[w] = y[w] = ~0ull;
                }
                if (a.out_a) store_kmer<N>(a.out_a, g, x);
                if constexpr (MODE == MODE_FW) {
                    if (a.out_b) store_kmer<N>(a.out_b, g, y);
                } else {
                    if (a.out_b) a.out_b[g] = y[0];
                }
            }
        };
        if constexpr (N == 1 && VEC && COVERED) {
            if (a.stride != 1u) {
                one_by_one();
                return;
            }
            // One-word kmers: a lane takes RUN = 4 consecutive elements per pass.  They usually belong to one
            // record, and then only the first is cut out of the stream; the others follow by the reference's own
            // rolling step (shift_encoding / shift_first_encoding of the complement, CanonicalKmers.jl:131-144)
            // with the entering symbols taken from the words already loaded -- the gather is ALU-bound
            // otherwise (about 90 instructions per element against 45 this way).  Two 16-byte stores per
            // array per lane and pass, issued at once: with the windows in LDS nothing later waits behind them.
            constexpr uint32_t RUN = RG_RUN;
            const uint32_t top = (uint32_t)DST * (k - 1u);
            const uint64_t kbits = k >= 64u ? ~0ull : (1ull << k) - 1ull;
            uint32_t r = 0;
            for (uint32_t e = RUN * tid; e < a.tile; e += (uint32_t)RG_PASS) {
                const uint64_t g = e0 + e;
                if (g > e_last) break;
                uint64_t X[RUN] = {}, Y[RUN] = {};
                const uint32_t cnt = e_last - g + 1 < (uint64_t)RUN ? (uint32_t)(e_last - g + 1) : RUN;
                r = slot_of(g, r);
                uint64_t fw[1], rc[1], x[1], y[1];
                // The run lies in record slot r (sub-run A, lenA elements) and, when it crosses a boundary, goes on in the next
                // record that owns something (sub-run B).  Each sub-run is ONE window cut and rolling steps.  A wavefront of 64
                // runs over 95-kmer reads almost always has a lane that crosses (1 - 0.968^64 = 87 %): sending that lane through
                // the element-by-element path made every wavefront pay for it (120 instructions per element, measured; this
                // kernel is bound by them).  Only a run that touches three records (records shorter than RUN kmers) still does.
                const uint64_t offA = off_l[r + 1u];
                const uint32_t lenA = offA - g < (uint64_t)cnt ? (uint32_t)(offA - g) : cnt;
                const bool two = lenA < cnt;
                uint32_t rB = r + 1u;
                bool three = false;
                if (two) {
                    while (off_l[rB + 1u] <= g + lenA) ++rB;       // (records that own nothing are stepped over)
                    three = g + cnt - 1u >= off_l[rB + 1u];
                }
                if (!three) {
                    // window cut of the sub-run of `len` elements that starts at element gg of record slot rr
                    auto cut = [&](uint64_t gg, uint32_t rr, uint32_t len, uint64_t &f_out, uint64_t &r_out, uint32_t &s_out, uint64_t &fb_out) {
                        const uint64_t p = delta_l[rr] + gg;
                        const uint32_t span = k + len - 1u;        // symbols the sub-run reads
                        fb_out = 0;                                // flagged symbols of the sub-run (skip mode)
                        const uint64_t bit = p * (uint64_t)DST;
                        const uint64_t q = bit >> 6;
                        const uint32_t sh = (uint32_t)(bit & 63u);
                        const uint32_t need = (sh + (uint32_t)DST * span + 63u) >> 6;  // 1..3 stream words
                        const uint64_t rel = q - d.q_lo;                               // (wraps for a window before the stretch)
                        const bool staged = rel < (uint64_t)d.n_words && rel + need <= (uint64_t)d.n_words;
                        if (flags) {
                            const uint64_t fq = p >> 6;
                            const uint32_t fs = (uint32_t)(p & 63u);
                            uint64_t f0, f1 = 0;
                            if (staged) {
                                const uint32_t fr = (uint32_t)(fq - d.f_lo);
                                f0 = flg_l[fr];
                                if (fs + span > 64u) f1 = flg_l[fr + 1u];
                            } else {
                                f0 = flags[fq];
                                if (fs + span > 64u) f1 = flags[fq + 1];
                            }
                            uint64_t f = (f0 >> fs) | ((f1 << 1) << (63u - fs));
                            f &= (1ull << span) - 1ull;               // span <= 32 + RUN - 1
                            if (f && !a.skip) {                        // the first element whose window holds a flagged symbol
                                const uint32_t first = (uint32_t)__builtin_ctzll(f);
                                atomicMin(a.err_slot, (unsigned long long)(gg + (first >= k ? first - k + 1u : 0u)));
                            }
                            if (a.skip) fb_out = f;
                        }
                        uint64_t l0, l1 = 0, l2 = 0;
                        if (staged) {
                            l0 = src_l[(uint32_t)rel];
                            if (need > 1u) l1 = src_l[(uint32_t)rel + 1u];
                            if (need > 2u) l2 = src_l[(uint32_t)rel + 2u];
                        } else {
                            l0 = a.stream[q];
                            if (need > 1u) l1 = a.stream[q + 1];
                            if (need > 2u) l2 = a.stream[q + 2];
                        }
                        const uint64_t W0 = funnel64(l0, l1, sh), W1 = funnel64(l1, l2, sh);
                        f_out = rev_symbols<DST>(W0 & mask) >> (64u - (uint32_t)DST * k);
                        r_out = comp_symbols<DST>(W0 & mask);
                        if constexpr (DST == 2) r_out &= mask;
                        // symbols K, K+1, ... of the sub-run
                        s_out = (uint32_t)((uint32_t)DST * k == 64u ? W1 : funnel64(W0, W1, (uint32_t)DST * k));
                    };
                    uint64_t fwB = 0, rcB = 0, fbits, fbB = 0;
                    uint32_t S, SB = 0;
                    cut(g, r, lenA, fw[0], rc[0], S, fbits);
                    if (two) cut(g + lenA, rB, cnt - lenA, fwB, rcB, SB, fbB);
                    uint32_t base = 0;                             // run index of the current sub-run's first element
                    finish(fw, rc, x, y);
                    X[0] = (fbits & kbits) ? ~0ull : x[0];
                    Y[0] = (fbits & kbits) ? ~0ull : y[0];
#pragma unroll
                    for (uint32_t j = 1; j < RUN; ++j) {
                        if (j < cnt) {
                            if (two && j == lenA) {                // the next record's first element: its own cut
                                fw[0] = fwB;
                                rc[0] = rcB;
                                S = SB;
                                fbits = fbB;
                                base = lenA;
                            } else {
                                const uint64_t sym = (S >> ((uint32_t)DST * (j - base - 1u))) & ((1u << DST) - 1u);
                                uint64_t csym;
                                if constexpr (DST == 2) csym = sym ^ 3u;
                                else csym = ((sym & 1u) << 3) | ((sym & 2u) << 1) | ((sym & 4u) >> 1) | ((sym & 8u) >> 3);
                                fw[0] = ((fw[0] << DST) | sym) & mask;
                                rc[0] = (rc[0] >> DST) | (csym << top);
                            }
                            finish(fw, rc, x, y);
                            const bool masked = ((fbits >> (j - base)) & kbits) != 0;
                            X[j] = masked ? ~0ull : x[0];
                            Y[j] = masked ? ~0ull : y[0];
                        }
                    }
                } else {                                           // the run touches three or more records
                    uint32_t rj = r;
#pragma unroll
                    for (uint32_t j = 0; j < RUN; ++j) {
                        if (j < cnt) {
                            while (off_l[rj + 1u] <= g + j) ++rj;  // (records that own nothing are stepped over)
                            const bool masked = element(g + j, rj, fw, rc);
                            finish(fw, rc, x, y);
                            X[j] = masked ? ~0ull : x[0];
                            Y[j] = masked ? ~0ull : y[0];
                        }
                    }
                }
                if (g + RUN - 1 <= e_last) {
                    if (a.out_a) {
#pragma unroll
                        for (uint32_t j = 0; j < RUN; j += 2) *reinterpret_cast<ulonglong2 *>(a.out_a + g + j) = make_ulonglong2(X[j], X[j + 1]);
                    }
                    if (a.out_b) {
#pragma unroll
                        for (uint32_t j = 0; j < RUN; j += 2) *reinterpret_cast<ulonglong2 *>(a.out_b + g + j) = make_ulonglong2(Y[j], Y[j + 1]);
                    }
                } else {
                    for (uint32_t j = 0; j < RUN && g + j <= e_last; ++j) {
                        if (a.out_a) a.out_a[g + j] = X[j];
                        if (a.out_b) a.out_b[g + j] = Y[j];
                    }
                }
            }
            RG_PROBE(4);
        } else {
            one_by_one();
        }
    };
    if (covered) run(std::integral_constant<bool, true>{});
    else run(std::integral_constant<bool, false>{});
    RG_PROBE(5);
#ifdef KMERS_RG_PROBE
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0): the stores have left
    RG_PROBE(6);
#endif
}

// Kmers of more than four words (Kmer{A,K,N} has no bound on N): one lane per element, the width a run-time argument
// (wide_kernel.hpp), the record found by a search of the global offsets, every word cut out of the stream in HBM.
// Same results and the same error rule as ragged_kernel; an edge path, not tuned.
template <int DST, int MODE>
__global__ __launch_bounds__(256) void ragged_wide_kernel(const RaggedArgs a, const uint32_t n_words) {
    const uint64_t g = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (g >= a.n_elems) return;
    uint64_t lo = 0, hi = a.n_records;  // the LAST record with off <= g (records that own nothing share their offset with the next)
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (a.rec_off[mid] <= g) lo = mid;
        else hi = mid;
    }
    const uint64_t p = a.spans[lo].first_base + (g - a.rec_off[lo]) * a.stride + a.stream_origin;
    bool bad = false;
    if (a.flags && *a.any_flag) {
        const uint64_t fq = p >> 6;
        bad = any_flag_in([&](uint32_t i) { return a.flags[fq + i]; }, (uint32_t)(p & 63u), a.k);
        if (bad && !a.skip) atomicMin(a.err_slot, (unsigned long long)g);
    }
    const bool masked = bad && a.skip;  // the all-ones sentinel
    auto load = [&](uint64_t q) -> uint64_t { return a.stream[q]; };  // (the stream is the kmer alphabet's symbols already)
    if constexpr (MODE == MODE_FW) {
        for (uint32_t w = 0; w < n_words; ++w) {
            if (a.out_a) a.out_a[g * n_words + w] = masked ? ~0ull : wide_word_from_stream<DST>(load, p, a.k, n_words, w, false);
            if (a.out_b) a.out_b[g * n_words + w] = masked ? ~0ull : wide_word_from_stream<DST>(load, p, a.k, n_words, w, true);
        }
    } else {
        const bool take_fw = wide_forward_is_canonical_from_stream<DST>(load, p, a.k, n_words);
        uint64_t h = a.seed;
        for (uint32_t w = 0; w < n_words; ++w) {
            const uint64_t c = wide_word_from_stream<DST>(load, p, a.k, n_words, w, !take_fw);
            if (a.out_a) a.out_a[g * n_words + w] = masked ? ~0ull : c;
            h = fx_step(h, c);
        }
        if (a.out_b) a.out_b[g] = masked ? ~0ull : h;
    }
}

}  // namespace kmers
