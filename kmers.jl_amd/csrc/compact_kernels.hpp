// compact_kernels.hpp -- UnambiguousKmers (src/iterators/UnambiguousKmers.jl:59-148): every
// window of K unambiguous symbols together with its 1-based start index.
//
// The reference walks the sequence with a `remaining` counter that is reset by every
// ambiguous symbol (:140-146).  Here a window is emitted iff the K bits of a one-bit-per-
// symbol ambiguity stream (staged in LDS next to the 2-bit code stream) are all zero, which
// selects exactly the same windows.  The output count is data dependent, so the work is
// count (bit-parallel, 64 starts per lane) -> exclusive scan -> emit, with the order of the
// reference preserved: in the emit pass each wavefront owns a contiguous quarter of its tile
// and compacts with ballot + popcount.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

struct CompactArgs {
    const uint64_t *src;
    uint64_t first_bit;
    uint64_t n_cand;          // candidate starts = n_bases - K + 1
    uint64_t n_tiles;
    uint32_t *counts;         // [n_tiles * WAVES] per (tile, wave)
    const uint64_t *offsets;  // exclusive scan of counts
    uint64_t *out_kmers;      // nullable
    long long *out_starts;    // nullable
    uint64_t index_origin;
    uint32_t k;
    uint32_t stride;          // keep windows with (start0 % stride) == 0
    uint32_t tile_kmers;      // multiple of 256
    uint32_t ascii_table;     // SRC_BITS == 8: ASCII_TABLE_SKIPPING = the reference's ASCII_SKIPPING_LUT (common.jl:22-32)
    unsigned long long *err_slot;  // SRC_BITS == 8: first invalid byte (0xff in the table) -> EncodeError
    uint64_t n_bases;         // SRC_BITS == 8: every byte below n_bases is inspected (UnambiguousKmers.jl:117-123)
    uint32_t tuples;          // 1: out_kmers receives Tuple{Kmer,Int} elements (N + 1 words each), out_starts unused
    uint32_t group;           // tiles staged together per workgroup iteration (count pass: > 1, emit pass: 1)
    uint32_t vec16;           // emit pass: out_kmers / out_starts are 16-byte aligned (16-byte stores allowed)
    uint64_t *xor_out;        // XOR instantiation: the accumulator
    uint32_t slot_mult;       // emit pass: its tile spans slot_mult tiles of the count pass (offsets are per count slot)
};

// DENSE: compiled with the no-compaction fast path for wavefronts whose starts are all kept (the host
// picks it when at least 90 % of the starts survive; the extra code costs the sparse case 10 %)
// XOR: no output arrays, no offsets: the head words of the kept kmers are XOR-folded into *xor_out (the reducer of
// test/benchmark.jl:9-15 over UnambiguousKmers: `y ⊻= first(x).data[1]`)
template <int SRC_BITS, int N, bool DENSE = false, bool XOR = false>
__global__ __launch_bounds__(BLOCK, 8) void unambiguous_kernel(const CompactArgs a) {  // 8 waves per SIMD: at most 64 VGPRs (65 without)
    __shared__ uint64_t lds[LDS_QWORDS];
    __shared__ uint64_t amb[MAX_TILE_BASES / 64 + 8];
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    constexpr uint32_t KEEP_ROUND = 1024;                       // starts per wavefront and compaction round (16 per lane)
    __shared__ uint16_t kept[XOR ? 1 : WAVES * KEEP_ROUND];     // per wavefront: the kept starts of a round
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += BLOCK) lut[i] = ascii_entry(a.ascii_table, i);
    }
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, 2);
    const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    uint64_t xacc = 0;  // XOR instantiation only

    // A workgroup iteration stages `group` consecutive tiles at once; the emit pass runs with
    // group = 1 (short-lived workgroups, the output stream sets the pace).
    const uint32_t group = a.group;
    const uint64_t n_groups = (a.n_tiles + group - 1) / group;
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t m0 = grp * group * a.tile_kmers;
        const uint64_t left = a.n_cand - m0;
        const uint32_t span = group * a.tile_kmers;
        const uint32_t mt = left < span ? (uint32_t)left : span;
        const uint64_t bit0 = a.first_bit + m0 * SRC_BITS;
        const uint64_t w0 = bit0 >> 6;
        const uint32_t b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(mt - 1) + k) * SRC_BITS;
        const uint32_t nw = (uint32_t)(((end_bit + 63) >> 6) - w0);

        // this wavefront's output offset of the group's first tile: requested now so that its latency
        // overlaps the source loads (the emit pass runs with group == 1)
        const uint64_t tile0 = grp * group;
        uint64_t pos_first = 0;
        if constexpr (!XOR) pos_first = a.offsets[(tile0 * WAVES + wave) * a.slot_mult];
        block_sync();
        for (uint32_t wi = tid; wi < nw; wi += BLOCK) {
            uint64_t x = a.src[w0 + wi];
            if constexpr (SRC_BITS == 8) {
                uint32_t codes = 0, flags = 0;
                uint64_t f = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint32_t v = lut[(x >> (8 * j)) & 0xffu];
                    codes |= (v & 3u) << (2 * j);
                    flags |= (v >= 0xf0u ? 1u : 0u) << j;             // 0xf0: ambiguous -> skip the window
                    f |= (uint64_t)(v == 0xffu ? 1u : 0u) << (8 * j);  // 0xff: not a nucleotide -> throw
                }
                reinterpret_cast<uint16_t *>(lds)[wi] = (uint16_t)codes;
                reinterpret_cast<uint8_t *>(amb)[wi] = (uint8_t)flags;
                if (f) report_bad_symbols<8, true>(a.err_slot, a.first_bit, a.n_bases, 1u, k, w0 + wi, f, x, a.index_origin);
            } else if constexpr (SRC_BITS == 4) {
                uint64_t bad;
                uint32_t c = pack_4to2(x, bad);
                reinterpret_cast<uint32_t *>(lds)[wi] = c;
                reinterpret_cast<uint16_t *>(amb)[wi] = (uint16_t)bad_bits16(bad);
            } else {
                lds[wi] = x;
                reinterpret_cast<uint32_t *>(amb)[wi] = 0u;  // 32 symbols per 2-bit word, none ambiguous
            }
        }
        block_sync();

        for (uint32_t j = 0; j < group; ++j) {
            const uint64_t tile = grp * group + j;
            if (tile >= a.n_tiles) break;
            const uint32_t per_wave = a.tile_kmers / WAVES;
            const uint32_t r_begin = j * a.tile_kmers + wave * per_wave;
            const uint32_t r_end = r_begin + per_wave < mt ? r_begin + per_wave : mt;  // r_begin may exceed mt in the last tile
            uint64_t pos = pos_first;
            if constexpr (!XOR) {
                if (j != 0) pos = a.offsets[(tile * WAVES + wave) * a.slot_mult];
            }
            // only starts on the stride lattice are candidates: the first one at or after r_begin
            // is r_first, then every `stride`-th (stride 1: every start)
            uint32_t r_first = r_begin;
            if (a.stride > 1) {
                const uint64_t rem = (m0 + r_begin) % a.stride;  // wave-uniform, once per (tile, wave)
                r_first = r_begin + (rem ? (uint32_t)(a.stride - rem) : 0u);
            }
            const uint32_t n_lat = r_first < r_end ? (r_end - r_first + a.stride - 1u) / a.stride : 0u;
            if constexpr (N == 1 && DENSE) {
                // Dense wavefront (every one of its starts is kept -- the normal state of real sequence
                // outside its N blocks): no compaction; two consecutive kmers per lane (the second by the
                // rolling step) and 16-byte stores aligned to the parity of the output position.
                bool dense = false;
                if (a.stride == 1 && a.vec16 && !a.tuples && n_lat) {
                    // dense <=> no ambiguity flag over the symbols its windows cover (wave-uniform test in LDS,
                    // so that the output offset is not needed before the stores)
                    const uint32_t lo = r_first + b0, hi = r_first + n_lat - 1u + k + b0;  // symbol range [lo, hi)
                    const uint32_t q = (lo >> 6) + lane;
                    uint64_t word = 0;
                    if (q <= ((hi - 1u) >> 6)) {
                        word = amb[q];
                        if (q == (lo >> 6)) word &= ~0ull << (lo & 63u);
                        if (q == ((hi - 1u) >> 6) && (hi & 63u)) word &= (1ull << (hi & 63u)) - 1ull;
                    }
                    dense = __ballot(word != 0) == 0;
                }
                if (dense) {
                    const uint64_t origin = m0 + 1 + a.index_origin;  // start of candidate r is origin + r
                    const uint32_t head = (uint32_t)(pos & 1u);
                    auto single = [&](uint32_t e) {
                        uint64_t fw[1], rc[1];
                        window<1, 2>(lds, 2u * (r_first + e + b0), k, mask, fw, rc);
                        if (a.out_kmers) a.out_kmers[pos + e] = fw[0];
                        if (a.out_starts) a.out_starts[pos + e] = (long long)(origin + r_first + e);
                    };
                    if (head && lane == 0) single(0);
                    const uint32_t pairs = (n_lat - head) >> 1;
                    for (uint32_t i = lane; i < pairs; i += 64u) {
                        const uint32_t e = head + 2u * i;
                        uint64_t f0, r0, sym;
                        window1_and_next<2>(lds, 2u * (r_first + e + b0), k, mask, f0, r0, sym);
                        const uint64_t f1 = ((f0 << 2) | sym) & mask;
                        if (a.out_kmers) *reinterpret_cast<ulonglong2 *>(a.out_kmers + pos + e) = make_ulonglong2(f0, f1);
                        if (a.out_starts)
                            *reinterpret_cast<ulonglong2 *>(a.out_starts + pos + e) =
                                make_ulonglong2(origin + r_first + e, origin + r_first + e + 1);
                    }
                    if (((n_lat - head) & 1u) && lane == 0) single(n_lat - 1);
                    continue;
                }
            }
            const uint32_t passes = (n_lat + 63u) / 64u;  // wave-uniform
            // is candidate li (of this wavefront's chunk of the tile) a window of K unambiguous symbols?
            auto kept_at = [&](uint32_t li) -> bool {
                const uint32_t bit = r_first + li * a.stride + b0;
                return (funnel64(amb[bit >> 6], amb[(bit >> 6) + 1], bit & 63u) & kmask) == 0;
            };
            if constexpr (XOR) {
                for (uint32_t p = 0; p < passes; ++p) {
                    const uint32_t li = p * 64u + lane;
                    if (li < n_lat && kept_at(li)) {
                        uint64_t fw[N], rc[N];
                        window<N, 2>(lds, 2u * (r_first + li * a.stride + b0), k, mask, fw, rc);
                        xacc ^= fw[0];
                    }
                }
            } else {
                // Survivors are sparse here (the dense case left above).  Round by round (up to 1024 starts of this
                // wavefront's chunk of the tile): every lane resolves its slice of consecutive starts bit-parallel, as the
                // count pass does (good bits AND-ed with themselves shifted by 1, 2, 4, ...: bit j survives iff K
                // unambiguous symbols begin there), the kept starts are listed in LDS in the reference's order (prefix
                // sum of the popcounts), and the list is worked off with every lane busy and the stores of a wavefront
                // contiguous.  (Testing start by start with a ballot per 64 cost three times the instructions: the
                // kernel is bound by its integer work.)
                uint16_t *mine = kept + wave * KEEP_ROUND;
                auto emit_listed = [&](uint32_t cnt) {  // list entries = starts relative to r_begin
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    for (uint32_t i = lane; i < cnt; i += 64u) {
                        const uint32_t r = r_begin + (uint32_t)mine[i];
                        const uint64_t g = m0 + r;
                        uint64_t fw[N], rc[N];
                        window<N, 2>(lds, 2u * (r + b0), k, mask, fw, rc);
                        const uint64_t o2 = pos + i;
                        if (a.tuples) {  // Tuple{Kmer,Int}: eltype of UnambiguousKmers (UnambiguousKmers.jl:39-41)
#pragma unroll
                            for (int wd = 0; wd < N; ++wd) a.out_kmers[o2 * (N + 1) + wd] = fw[wd];
                            a.out_kmers[o2 * (N + 1) + N] = g + 1 + a.index_origin;
                        } else {
                            if (a.out_kmers) {
#pragma unroll
                                for (int wd = 0; wd < N; ++wd) a.out_kmers[o2 * N + wd] = fw[wd];
                            }
                            if (a.out_starts) a.out_starts[o2] = (long long)(g + 1 + a.index_origin);
                        }
                    }
                    pos += cnt;
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();  // the list is rewritten by the next round
                };
                if (a.stride > 1) {
                    // a stride lattice: only every stride-th start is a candidate; those are tested one per lane (ballot +
                    // popcount, four rounds of 64 per list)
                    for (uint32_t p0 = 0; p0 < passes; p0 += 4u) {
                        uint32_t cnt = 0;
                        const uint32_t p1 = p0 + 4u < passes ? p0 + 4u : passes;
                        for (uint32_t p = p0; p < p1; ++p) {  // the whole wave iterates together (ballot needs every lane)
                            const uint32_t li = p * 64u + lane;
                            const bool ok = li < n_lat && kept_at(li);
                            const uint64_t bal = __ballot(ok);
                            if (ok) mine[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(r_first + li * a.stride - r_begin);
                            cnt += __popcll(bal);
                        }
                        emit_listed(cnt);
                    }
                    continue;
                }
                for (uint32_t done = r_begin; done < r_end; done += KEEP_ROUND) {
                    const uint32_t n_round = r_end - done < KEEP_ROUND ? r_end - done : KEEP_ROUND;
                    const uint32_t w = (n_round + 63u) / 64u;         // starts per lane: 1 .. KEEP_ROUND / 64
                    const uint32_t s0 = done + lane * w;              // this lane's first start (tile-relative)
                    uint64_t keep = 0;
                    if (s0 < done + n_round) {
                        const uint32_t bit = s0 + b0;
                        const uint32_t Q = bit >> 6, sft = bit & 63u;
                        uint64_t lo = ~funnel64(amb[Q], amb[Q + 1], sft), hi = ~funnel64(amb[Q + 1], amb[Q + 2], sft);
                        uint32_t have = 1;
                        while (have < k) {
                            const uint32_t step = have < k - have ? have : k - have;   // 1..63
                            lo &= (lo >> step) | ((hi << 1) << (63u - step));
                            hi &= hi >> step;
                            have += step;
                        }
                        const uint32_t valid = done + n_round - s0 < w ? done + n_round - s0 : w;  // starts of the slice that exist
                        keep = lo & ((1ull << valid) - 1ull);         // w <= 16
                    }
                    const uint32_t c = (uint32_t)__popcll(keep);
                    uint32_t incl = c;
                    for (int d = 1; d < 64; d <<= 1) {
                        const uint32_t y = __shfl_up(incl, d, 64);
                        if ((int)lane >= d) incl += y;
                    }
                    const uint32_t cnt = __shfl(incl, 63, 64);
                    uint32_t o = incl - c;
                    while (keep) {
                        mine[o++] = (uint16_t)(s0 - r_begin + (uint32_t)__builtin_ctzll(keep));
                        keep &= keep - 1ull;
                    }
                    emit_listed(cnt);
                }
            }
        }
    }
    if constexpr (XOR) {
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if (lane == 0) atomicXor(reinterpret_cast<unsigned long long *>(a.xor_out), (unsigned long long)xacc);
    }
}

// Count pass, bit-parallel: one lane resolves 64 candidate starts at once.  With g = the "good"
// (unambiguous) bit of every symbol, start i is kept iff g[i..i+K) are all ones; AND-ing the
// stream with itself shifted by 1, 2, 4, ... (binary decomposition of K) leaves exactly those
// bits.  Counts are accumulated per (tile, wavefront) of the EMIT pass, which owns the layout.
template <int SRC_BITS>
__global__ __launch_bounds__(BLOCK) void unambiguous_count_kernel(const CompactArgs a) {
    __shared__ uint64_t amb[MAX_TILE_BASES / 64 + 8];
    __shared__ uint32_t cnt[MAX_TILE_BASES / 64];          // one counter per (tile, wave) of the group (>= 64 starts each)
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint32_t per_wave = a.tile_kmers / WAVES;        // starts per (tile, wave), a multiple of 64
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += BLOCK) lut[i] = ascii_entry(a.ascii_table, i);
    }
    const uint32_t group_starts = a.group * a.tile_kmers;  // whole tiles, <= MAX_TILE_BASES starts per iteration
    const uint64_t n_groups = (a.n_cand + group_starts - 1) / group_starts;
    for (uint64_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const uint64_t m0 = grp * group_starts;            // first candidate start of the group (tile aligned)
        const uint64_t left = a.n_cand - m0;
        const uint32_t mt = left < group_starts ? (uint32_t)left : group_starts;
        const uint64_t bit0 = a.first_bit + m0 * SRC_BITS;
        const uint64_t w0 = bit0 >> 6;
        const uint32_t b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(mt - 1) + k) * SRC_BITS;
        const uint32_t nw = (uint32_t)(((end_bit + 63) >> 6) - w0);
        const uint32_t n_slots = (mt + per_wave - 1) / per_wave;

        block_sync();
        for (uint32_t i = tid; i < n_slots; i += BLOCK) cnt[i] = 0;
        for (uint32_t wi = tid; wi < nw; wi += BLOCK) {
            uint64_t x = a.src[w0 + wi];
            if constexpr (SRC_BITS == 8) {
                uint32_t flags = 0;
                uint64_t f = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    uint32_t v = lut[(x >> (8 * j)) & 0xffu];
                    flags |= (v >= 0xf0u ? 1u : 0u) << j;
                    f |= (uint64_t)(v == 0xffu ? 1u : 0u) << (8 * j);
                }
                reinterpret_cast<uint8_t *>(amb)[wi] = (uint8_t)flags;
                if (f) report_bad_symbols<8, true>(a.err_slot, a.first_bit, a.n_bases, 1u, k, w0 + wi, f, x, a.index_origin);
            } else if constexpr (SRC_BITS == 4) {
                uint64_t bad;
                (void)pack_4to2(x, bad);
                reinterpret_cast<uint16_t *>(amb)[wi] = (uint16_t)bad_bits16(bad);
            } else {
                reinterpret_cast<uint32_t *>(amb)[wi] = 0u;
            }
        }
        // (flag bits past the staged range only ever reach starts >= mt, which are masked out below)
        block_sync();

        const uint32_t n_q = (mt + 63u) / 64u;             // 64 starts per lane
        for (uint32_t q = tid; q < n_q; q += BLOCK) {
            const uint32_t bit = 64u * q + b0;
            const uint32_t Q = bit >> 6, sft = bit & 63u;
            uint64_t lo = ~funnel64(amb[Q], amb[Q + 1], sft), hi = ~funnel64(amb[Q + 1], amb[Q + 2], sft);
            uint32_t have = 1;
            while (have < k) {
                const uint32_t step = have < k - have ? have : k - have;   // 1..63
                const uint64_t slo = (lo >> step) | ((hi << 1) << (63u - step));
                const uint64_t shi = hi >> step;
                lo &= slo;
                hi &= shi;
                have += step;
            }
            uint64_t keep = lo;                            // bit j: start 64q + j begins K unambiguous symbols
            const uint32_t valid = mt - 64u * q;           // starts of this qword that exist
            if (valid < 64u) keep &= (1ull << valid) - 1ull;
            if (a.stride > 1) {                            // keep only starts with (m0 + 64q + j) % stride == 0
                const uint64_t rem = (m0 + 64ull * q) % a.stride;
                uint64_t lat = 0;
                for (uint64_t pbit = rem ? a.stride - rem : 0; pbit < 64; pbit += a.stride) lat |= 1ull << pbit;
                keep &= lat;
            }
            const uint32_t c = (uint32_t)__popcll(keep);
            if (c) atomicAdd(&cnt[(64u * q) / per_wave], c);
        }
        block_sync();
        // slot i of the group = (tile, wave) number (m0 / per_wave + i) of the emit pass
        for (uint32_t i = tid; i < n_slots; i += BLOCK) a.counts[m0 / per_wave + i] = cnt[i];
    }
}

// ---- exclusive scan of n 32-bit counts into 64-bit offsets (offsets[n] = total) -------------
// Three coalesced kernels: per-segment sums -> scan of the segment sums (one workgroup) ->
// per-segment rescan with the segment's base.  A segment is SCAN_SEG consecutive counts.
constexpr uint32_t SCAN_SEG = 2048;  // 256 threads x 8

__device__ __forceinline__ uint64_t block_reduce_sum(uint64_t v, uint64_t *tmp) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0) tmp[threadIdx.x >> 6] = v;
    block_sync();
    uint64_t total = 0;
    for (uint32_t w = 0; w < blockDim.x / 64; ++w) total += tmp[w];
    block_sync();
    return total;
}

__global__ __launch_bounds__(256) void scan_segment_sums_kernel(const uint32_t *__restrict__ counts, uint64_t n,
                                                                 uint64_t *__restrict__ seg_sums) {
    __shared__ uint64_t tmp[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_SEG;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) {
        uint64_t i = base + threadIdx.x + 256u * j;
        if (i < n) v += counts[i];
    }
    uint64_t total = block_reduce_sum(v, tmp);
    if (threadIdx.x == 0) seg_sums[blockIdx.x] = total;
}

// in-place exclusive scan of the segment sums; seg_sums[n_seg] = grand total
__global__ __launch_bounds__(1024) void scan_segments_kernel(uint64_t *__restrict__ seg_sums, uint64_t n_seg) {
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint64_t chunk = (n_seg + 1023) / 1024;
    const uint64_t lo = (uint64_t)t * chunk < n_seg ? (uint64_t)t * chunk : n_seg, hi = lo + chunk < n_seg ? lo + chunk : n_seg;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; ++i) s += seg_sums[i];
    part[t] = s;
    block_sync();
    for (uint32_t d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan of the partials
        uint64_t v = t >= d ? part[t - d] : 0;
        block_sync();
        part[t] += v;
        block_sync();
    }
    uint64_t run = t ? part[t - 1] : 0;
    for (uint64_t i = lo; i < hi; ++i) {
        uint64_t c = seg_sums[i];
        seg_sums[i] = run;
        run += c;
    }
    if (t == 1023) seg_sums[n_seg] = part[1023];
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const uint32_t *__restrict__ counts, uint64_t n,
                                                          const uint64_t *__restrict__ seg_sums, uint64_t n_seg,
                                                          uint64_t *__restrict__ offsets) {
    __shared__ uint32_t c[SCAN_SEG];
    __shared__ uint64_t wave_tot[4];
    const uint32_t t = threadIdx.x;
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_SEG;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) {
        uint64_t i = base + t + 256u * j;
        c[t + 256u * j] = i < n ? counts[i] : 0u;
    }
    block_sync();
    // thread t owns 8 consecutive counts: local exclusive prefix, then a scan over the thread totals
    uint64_t local[SCAN_SEG / 256];  // 64-bit: kmers_batch scans per-record counts of up to 2^32 - 1
    uint64_t sum = 0;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) {
        local[j] = sum;
        sum += c[t * (SCAN_SEG / 256) + j];
    }
    uint64_t incl = sum;
    const uint32_t lane = t & 63u, wave = t >> 6;
    for (int off = 1; off < 64; off <<= 1) {
        uint64_t v = __shfl_up(incl, off, 64);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    block_sync();
    uint64_t before = seg_sums[blockIdx.x];
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    const uint64_t excl = before + incl - sum;
#pragma unroll
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) {
        uint64_t i = base + (uint64_t)t * (SCAN_SEG / 256) + j;
        if (i < n) offsets[i] = excl + local[j];
    }
    if (blockIdx.x == 0 && t == 0) offsets[n] = seg_sums[n_seg];
}

}  // namespace kmers
