// ragged_kernels.hpp -- the iterators over a BATCH of records (reads, contigs, FASTA records)
// packed in one pool of source words.  The reference iterates one sequence at a time
// (`for record in reader; ... CanonicalDNAMers{K}(sequence(record)) ...`, docs/src/minhash.md:31-35);
// on the GPU a call costs ~16 us, so short records are only worth it many at a time: one launch
// produces the elements of all records, concatenated in record order.
//
//   1. recode_kernel   (skipped for Copyable 2->2 / 4->4 pools): the pool's symbols -> a DST-bit
//      stream in HBM + one "cannot be encoded" flag per symbol (RecodingScheme,
//      src/construction.jl:75-100).  Elementwise, r B/base read, (DST+1)/8 B/base written.
//   2. ragged_kernel: one tile of 1024 output elements per workgroup.  The host supplies, per tile,
//      the first record it touches; the tile's slice of (element offset, first symbol) pairs is
//      staged in LDS, every lane finds its record by binary search there, reads its window's stream
//      words straight from HBM (neighbouring lanes share them through L1/L2) and derives the kmer
//      exactly like `window()` of stream_kernel.hpp.  A window that covers a flagged symbol reports
//      its element index (atomicMin): the smallest one is the first element the reference would
//      have failed on, in record order, and its first flagged symbol the one it throws for.
#pragma once
#include <type_traits>

#include "stream_kernel.hpp"

namespace kmers {

#ifndef KMERS_RG_TILE
#define KMERS_RG_TILE 2048
#endif
constexpr int RG_TILE = KMERS_RG_TILE;  // output elements per workgroup
constexpr int RG_SLOTS = 256;            // records of a tile staged in LDS (more: the global-search path)

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
    const uint32_t *tile_rec;    // [n_tiles + 1] the record that owns the first element of every tile; [n_tiles] = n - 1
    uint64_t n_records;
    uint64_t n_elems;
    uint64_t *out_a;             // FW: forward kmers, CANON: canonical kmers
    uint64_t *out_b;             // FW: reverse complements (nullable), CANON: fx_hash (nullable)
    uint64_t seed;
    unsigned long long *err_slot;  // atomicMin of the first failing ELEMENT index
    uint32_t k;
    uint32_t skip;               // 1: elements whose window holds a flagged symbol are written as all-ones instead of failing
};

struct RecodeArgs {
    const uint64_t *src;         // pool words; word 0 holds pool symbol 0 at symbol offset (stream_origin)
    uint64_t n_words;
    uint64_t *stream;
    uint64_t *flags;
    uint64_t *any_flag;          // becomes non-zero if any symbol of the pool cannot be encoded
    uint32_t ascii_table;
};

// ---- the ragged layout, computed on the device (a batch may hold tens of millions of records) ------
// elements per record (FwKmers.jl:40-43) for the scan kernels of compact_kernels.hpp; bad[0] != 0 if a
// span reaches outside the pool or a record is too long for the 32-bit count
__global__ __launch_bounds__(256) void ragged_count_kernel(const RaggedSpan *__restrict__ spans, uint64_t n, uint32_t k,
                                                            uint64_t pool_bases, uint32_t *__restrict__ counts,
                                                            uint64_t *__restrict__ bad) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += stride) {
        const RaggedSpan sp = spans[i];
        if (sp.first_base > pool_bases || sp.n_bases > pool_bases - sp.first_base || sp.n_bases >= 0xFFFFFFFFull) {
            bad[0] = 1;
            counts[i] = 0;
        } else {
            counts[i] = sp.n_bases < k ? 0u : (uint32_t)(sp.n_bases - k + 1u);
        }
    }
}

// the record that owns element t * RG_TILE: the LAST i with off[i] <= that element (records that own
// nothing share their offset with the next one and are skipped by "last")
__global__ __launch_bounds__(256) void ragged_tiles_kernel(const uint64_t *__restrict__ off, uint64_t n, uint64_t n_tiles,
                                                            uint32_t *__restrict__ tile_rec) {
    const uint64_t t = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (t > n_tiles) return;
    if (t == n_tiles) {  // sentinel: lets the last tile size its slice like the others
        tile_rec[t] = (uint32_t)(n - 1);
        return;
    }
    const uint64_t e = t * RG_TILE;
    uint64_t lo = 0, hi = n;  // off[0] = 0 <= e
    while (hi - lo > 1) {
        const uint64_t mid = (lo + hi) >> 1;
        if (off[mid] <= e) lo = mid;
        else hi = mid;
    }
    tile_rec[t] = (uint32_t)lo;
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
    for (uint64_t wi = (uint64_t)blockIdx.x * 256u + threadIdx.x; wi < a.n_words; wi += stride) {
        const uint64_t x = a.src[wi];
        if constexpr (SRC == 4 && DST == 2) {
            uint64_t bad;
            reinterpret_cast<uint32_t *>(a.stream)[wi] = pack_4to2(x, bad);
            reinterpret_cast<uint16_t *>(a.flags)[wi] = (uint16_t)bad_bits16(bad);
            if (bad) *a.any_flag = 1;  // rare; any writer, same value
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

// `window()` of stream_kernel.hpp with the stream words coming from HBM
template <int N, int DST>
__device__ __forceinline__ void window_global(const uint64_t *__restrict__ stream, uint64_t bit, uint32_t k, uint64_t mask,
                                              uint64_t (&fw)[N], uint64_t (&rc)[N]) {
    const uint64_t q = bit >> 6;
    const uint32_t s = (uint32_t)(bit & 63u);
    uint64_t W[N], R[N];
    uint64_t lo = stream[q];
    const uint32_t need = (s + (uint32_t)DST * k + 63u) >> 6;  // stream words the window touches (<= N + 1)
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const uint64_t hi = (uint32_t)(j + 1) < need ? stream[q + j + 1] : 0;  // never read past the window's last word
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

// VEC: out_a / out_b are 16-byte aligned (one-word kmers: two elements per lane, 16-byte stores)
template <int DST, int N, int MODE, bool VEC>
__global__ __launch_bounds__(256) void ragged_kernel(const RaggedArgs a) {
    __shared__ uint64_t off_l[RG_SLOTS + 1];
    __shared__ uint64_t base_l[RG_SLOTS + 1];
    __shared__ __attribute__((aligned(16))) uint32_t owner[RG_TILE];
    __shared__ uint32_t wave_max[4];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, DST);
    // (the pool may hold flagged symbols outside every record, so a set any_flag only means "look")
    const uint64_t *flags = (a.flags && *a.any_flag) ? a.flags : nullptr;
    const uint64_t tile = blockIdx.x;
    const uint64_t e0 = tile * RG_TILE;
    const uint64_t r_lo = a.tile_rec[tile];
    const uint64_t e_last = (e0 + RG_TILE < a.n_elems ? e0 + RG_TILE : a.n_elems) - 1;
    // offsets and first symbols of this tile's records go to LDS: r_lo .. the record that owns the next
    // tile's first element, plus the one after it (its offset lies past this tile: the end of the
    // search range; off[n] = n_elems closes the last tile).  At most RG_SLOTS + 1 slots: a tile crowded with
    // records that own nothing is not covered, and its lanes search the global arrays instead
    // (correct, just slower).
    const uint64_t want = (uint64_t)a.tile_rec[tile + 1] - r_lo + 2;
    const uint32_t n_rec = want < (uint64_t)(RG_SLOTS + 1) ? (uint32_t)want : (uint32_t)(RG_SLOTS + 1);
    for (uint32_t i = tid; i < n_rec; i += 256u) {
        off_l[i] = a.rec_off[r_lo + i];
        base_l[i] = r_lo + i < a.n_records ? a.spans[r_lo + i].first_base : 0;
    }
    for (uint32_t i = tid; i < (uint32_t)RG_TILE; i += 256u) owner[i] = 0;
    block_sync();
    const bool covered = off_l[n_rec - 1] > e_last;
    if (covered) {
        // owner[e] = record slot of element e0 + e, without a search per element: every slot marks the
        // element it starts at (records that own nothing share a start with the next one: max wins),
        // then a running maximum over the tile fills the rest.
        for (uint32_t i = 1u + tid; i < n_rec; i += 256u) {
            const uint64_t pos = off_l[i] - e0;  // > 0: slot 0 is the last record with off <= e0
            if (pos < (uint64_t)RG_TILE) atomicMax(&owner[(uint32_t)pos], i);
        }
        block_sync();
        constexpr uint32_t PER = RG_TILE / 256;          // consecutive elements per thread
        uint32_t v[PER];
#pragma unroll
        for (uint32_t j = 0; j < PER; j += 4) {
            const uint4 q = reinterpret_cast<uint4 *>(owner)[tid * (PER / 4) + j / 4];
            v[j] = q.x; v[j + 1] = q.y; v[j + 2] = q.z; v[j + 3] = q.w;
        }
#pragma unroll
        for (uint32_t j = 1; j < PER; ++j) v[j] = max(v[j], v[j - 1]);
        uint32_t m = v[PER - 1];                         // inclusive maximum up to this thread's last element
        const uint32_t lane = tid & 63u, wave = tid >> 6;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(m, d, 64);
            if ((int)lane >= d) m = max(m, x);
        }
        if (lane == 63u) wave_max[wave] = m;
        uint32_t before = __shfl_up(m, 1, 64);           // maximum of the earlier lanes of this wavefront
        if (lane == 0) before = 0;
        block_sync();
        for (uint32_t w = 0; w < wave; ++w) before = max(before, wave_max[w]);
#pragma unroll
        for (uint32_t j = 0; j < PER; j += 4)
            reinterpret_cast<uint4 *>(owner)[tid * (PER / 4) + j / 4] =
                make_uint4(max(v[j], before), max(v[j + 1], before), max(v[j + 2], before), max(v[j + 3], before));
        // first stream symbol of a slot's record minus its element offset: window of element g = delta + g
        for (uint32_t i = tid; i < n_rec; i += 256u) base_l[i] = base_l[i] - off_l[i] + a.stream_origin;
        block_sync();
    }
    // Two separate instantiations of the element loop -- LDS lookups or global lookups -- rather than a
    // per-access select: hipcc turned `covered ? lds[i] : global[i]` into FLAT loads whose address is
    // selected between the LDS aperture and HBM, and that version produced wrong results for whole
    // wavefronts, non-deterministically, on gfx950 (tools/ history, r01_tuning.md).
    auto run = [&](auto covered_tag) {
        constexpr bool COVERED = decltype(covered_tag)::value;
        // record slot of element g: the last slot with off <= g (slot 0 always qualifies)
        auto record_of = [&](uint64_t g) -> uint64_t {
            if constexpr (COVERED) {
                return owner[(uint32_t)(g - e0)];
            } else {
                uint64_t lo = r_lo, hi = a.n_records;
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (a.rec_off[mid] <= g) lo = mid;
                    else hi = mid;
                }
                return lo - r_lo;
            }
        };
        auto slot_off = [&](uint64_t r) -> uint64_t {
            return a.rec_off[r_lo + r];
        };
        auto slot_base = [&](uint64_t r) -> uint64_t {
            return a.spans[r_lo + r].first_base;
        };
        // element g of record slot r: forward kmer and reverse complement (and the flag test)
        auto element = [&](uint64_t g, uint64_t r, uint64_t (&fw)[N], uint64_t (&rc)[N]) -> bool {
            bool bad = false;
            uint64_t p;  // stream symbol index of the window
            if constexpr (COVERED) p = base_l[r] + g;
            else p = slot_base(r) + (g - slot_off(r)) + a.stream_origin;
            if (flags) {
                const uint64_t fq = p >> 6;
                const uint32_t fs = (uint32_t)(p & 63u);
                uint64_t f = flags[fq] >> fs;
                if (fs + k > 64u) f |= (flags[fq + 1] << 1) << (63u - fs);
                if (k < 64u) f &= (1ull << k) - 1ull;
                bad = f != 0;
                if (bad && !a.skip) atomicMin(a.err_slot, (unsigned long long)g);
            }
            window_global<N, DST>(a.stream, p * (uint64_t)DST, k, mask, fw, rc);
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
        if constexpr (N == 1 && VEC && COVERED) {
            // One-word kmers: a lane takes RUN = 4 consecutive elements.  They usually belong to one record,
            // and then only the first is cut out of the stream; the others follow by the reference's own
            // rolling step (shift_encoding / shift_first_encoding of the complement, CanonicalKmers.jl:131-144)
            // with the entering symbols taken from the words already loaded -- the gather is ALU-bound
            // otherwise (about 90 instructions per element against 45 this way).  Two 16-byte stores per
            // array per lane.  Every pass is fetched before anything is stored.
            constexpr uint32_t RUN = 4, PASSES = RG_TILE / (256 * RUN);
            uint64_t X[PASSES][RUN], Y[PASSES][RUN];
            const uint32_t top = (uint32_t)DST * (k - 1u);
#pragma unroll
            for (uint32_t ps = 0; ps < PASSES; ++ps) {
                const uint32_t e = RUN * tid + 256u * RUN * ps;
                const uint64_t g = e0 + e;
#pragma unroll
                for (uint32_t j = 0; j < RUN; ++j) X[ps][j] = Y[ps][j] = 0;
                if (g > e_last) continue;
                const uint32_t cnt = e_last - g + 1 < (uint64_t)RUN ? (uint32_t)(e_last - g + 1) : RUN;
                const uint4 o4 = *reinterpret_cast<const uint4 *>(&owner[e]);
                const uint32_t o[RUN] = {o4.x, o4.y, o4.z, o4.w};
                bool same = true;
#pragma unroll
                for (uint32_t j = 1; j < RUN; ++j) same = same && (j >= cnt || o[j] == o[0]);
                uint64_t fw[1], rc[1], x[1], y[1];
                if (same) {
                    const uint64_t p = base_l[o[0]] + g;
                    const uint32_t span = k + cnt - 1u;            // symbols the run reads
                    uint64_t fbits = 0;                            // flagged symbols of the run (skip mode)
                    if (flags) {
                        const uint64_t fq = p >> 6;
                        const uint32_t fs = (uint32_t)(p & 63u);
                        uint64_t f = flags[fq] >> fs;
                        if (fs + span > 64u) f |= (flags[fq + 1] << 1) << (63u - fs);
                        f &= (1ull << span) - 1ull;               // span <= 32 + 3
                        if (f && !a.skip) {                        // the first element whose window holds a flagged symbol
                            const uint32_t first = (uint32_t)__builtin_ctzll(f);
                            atomicMin(a.err_slot, (unsigned long long)(g + (first >= k ? first - k + 1u : 0u)));
                        }
                        if (a.skip) fbits = f;
                    }
                    const uint64_t kbits = k >= 64u ? ~0ull : (1ull << k) - 1ull;
                    const uint64_t bit = p * (uint64_t)DST;
                    const uint64_t q = bit >> 6;
                    const uint32_t sh = (uint32_t)(bit & 63u);
                    const uint32_t need = (sh + (uint32_t)DST * span + 63u) >> 6;  // 1..3 stream words
                    const uint64_t l0 = a.stream[q], l1 = need > 1u ? a.stream[q + 1] : 0, l2 = need > 2u ? a.stream[q + 2] : 0;
                    const uint64_t W0 = funnel64(l0, l1, sh), W1 = funnel64(l1, l2, sh);
                    fw[0] = rev_symbols<DST>(W0 & mask) >> (64u - (uint32_t)DST * k);
                    rc[0] = comp_symbols<DST>(W0 & mask);
                    if constexpr (DST == 2) rc[0] &= mask;
                    // symbols K, K+1, K+2 of the run
                    const uint32_t S = (uint32_t)((uint32_t)DST * k == 64u ? W1 : funnel64(W0, W1, (uint32_t)DST * k));
                    finish(fw, rc, x, y);
                    X[ps][0] = (fbits & kbits) ? ~0ull : x[0];
                    Y[ps][0] = (fbits & kbits) ? ~0ull : y[0];
#pragma unroll
                    for (uint32_t j = 1; j < RUN; ++j) {
                        if (j < cnt) {
                            const uint64_t sym = (S >> ((uint32_t)DST * (j - 1u))) & ((1u << DST) - 1u);
                            uint64_t csym;
                            if constexpr (DST == 2) csym = sym ^ 3u;
                            else csym = ((sym & 1u) << 3) | ((sym & 2u) << 1) | ((sym & 4u) >> 1) | ((sym & 8u) >> 3);
                            fw[0] = ((fw[0] << DST) | sym) & mask;
                            rc[0] = (rc[0] >> DST) | (csym << top);
                            finish(fw, rc, x, y);
                            const bool masked = ((fbits >> j) & kbits) != 0;
                            X[ps][j] = masked ? ~0ull : x[0];
                            Y[ps][j] = masked ? ~0ull : y[0];
                        }
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < RUN; ++j) {
                        if (j < cnt) {
                            const bool masked = element(g + j, o[j], fw, rc);
                            finish(fw, rc, x, y);
                            X[ps][j] = masked ? ~0ull : x[0];
                            Y[ps][j] = masked ? ~0ull : y[0];
                        }
                    }
                }
            }
#pragma unroll
            for (uint32_t ps = 0; ps < PASSES; ++ps) {
                const uint64_t g = e0 + RUN * tid + 256u * RUN * ps;
                if (g + RUN - 1 <= e_last) {
                    if (a.out_a) {
                        *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(X[ps][0], X[ps][1]);
                        *reinterpret_cast<ulonglong2 *>(a.out_a + g + 2) = make_ulonglong2(X[ps][2], X[ps][3]);
                    }
                    if (a.out_b) {
                        *reinterpret_cast<ulonglong2 *>(a.out_b + g) = make_ulonglong2(Y[ps][0], Y[ps][1]);
                        *reinterpret_cast<ulonglong2 *>(a.out_b + g + 2) = make_ulonglong2(Y[ps][2], Y[ps][3]);
                    }
                } else {
                    for (uint32_t j = 0; j < RUN && g + j <= e_last; ++j) {
                        if (a.out_a) a.out_a[g + j] = X[ps][j];
                        if (a.out_b) a.out_b[g + j] = Y[ps][j];
                    }
                }
            }
        } else {
            for (uint32_t e = tid; e < (uint32_t)RG_TILE; e += 256u) {
                const uint64_t g = e0 + e;
                if (g > e_last) break;
                uint64_t fw[N], rc[N], x[N], y[N];
                const bool masked = element(g, record_of(g), fw, rc);
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
        }
    };
    if (covered) run(std::integral_constant<bool, true>{});
    else run(std::integral_constant<bool, false>{});
}

}  // namespace kmers
