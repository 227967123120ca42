// batch_kernels.hpp -- element-wise kernels over arrays of kmers and the synthetic input
// generator.  Reference: src/kmer.jl:255-261 (fx_hash), src/transformations.jl:1-41
// (reverse / complement / reverse_complement / canonical / iscanonical).
#pragma once
#include "device_bits.hpp"

namespace kmers {

// ---- fx_hash over n kmers of NW words ----------------------------------------------------
// Short-lived workgroups, 16-byte accesses (the same launch-shape lesson as the stream kernel:
// profiles/r01_tuning.md): one-word kmers are processed two per lane.
template <int NW>
__global__ __launch_bounds__(256) void fx_hash_kernel(const uint64_t *__restrict__ kmers, uint64_t n,
                                                       uint64_t seed, uint64_t *__restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    if constexpr (NW == 1) {
        const uint64_t pairs = n / 2;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
            ulonglong2 x = reinterpret_cast<const ulonglong2 *>(kmers)[i];
            reinterpret_cast<ulonglong2 *>(out)[i] = make_ulonglong2(fx_step(seed, x.x), fx_step(seed, x.y));
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = fx_step(seed, kmers[n - 1]);
        return;
    }
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = seed;
#pragma unroll
        for (int w = 0; w < NW; ++w) h = fx_step(h, kmers[i * NW + w]);
        out[i] = h;
    }
}

// generic width (any number of words): used for NW > 4
__global__ __launch_bounds__(256) void fx_hash_kernel_any(const uint64_t *__restrict__ kmers, int nw, uint64_t n,
                                                           uint64_t seed, uint64_t *__restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint64_t h = seed;
        for (int w = 0; w < nw; ++w) h = fx_step(h, kmers[i * nw + w]);
        out[i] = h;
    }
}

// ---- whole-kmer transforms ---------------------------------------------------------------
// reverse the order of the BITS-wide symbols of one word (BioSequences.reversebits)
template <int BITS>
__device__ __forceinline__ uint64_t reverse_symbols(uint64_t x) {
    uint64_t r = __brevll(x);
    r = ((r >> 1) & 0x5555555555555555ull) | ((r & 0x5555555555555555ull) << 1);
    if constexpr (BITS == 4) r = ((r >> 2) & 0x3333333333333333ull) | ((r & 0x3333333333333333ull) << 2);
    return r;
}

// complement_bitpar: 2-bit NOT; 4-bit = bit reversal inside every nibble
template <int BITS>
__device__ __forceinline__ uint64_t complement_word(uint64_t x) {
    if constexpr (BITS == 2) return ~x;
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
    return ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
}

template <int NW, int BITS>
__device__ __forceinline__ void kmer_complement(uint64_t (&d)[NW], uint64_t mask) {
#pragma unroll
    for (int w = 0; w < NW; ++w) d[w] = complement_word<BITS>(d[w]);
    if constexpr (BITS == 2) d[0] &= mask;  // transformations.jl:24 (the 4-bit method needs no mask, :12-13)
}

// transformations.jl:1-10: reversebits of every word, tuple reversed, right shift by bits_unused
template <int NW, int BITS>
__device__ __forceinline__ void kmer_reverse(uint64_t (&d)[NW], uint32_t bu) {
    uint64_t t[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) t[w] = reverse_symbols<BITS>(d[NW - 1 - w]);
#pragma unroll
    for (int w = NW - 1; w >= 0; --w) {
        uint64_t carry_in = w > 0 ? ((t[w - 1] << 1) << (63u - bu)) : 0ull;  // low bu bits of the word above
        d[w] = (t[w] >> bu) | carry_in;
    }
}

// one kmer through `op`; returns false when the result is a single flag/count in y[0]
template <int NW, int BITS>
__device__ __forceinline__ bool transform_one(int op, const uint64_t (&x)[NW], uint64_t (&y)[NW], uint64_t mask, uint32_t bu) {
#pragma unroll
    for (int w = 0; w < NW; ++w) y[w] = x[w];
    if (op == 6) {  // count(isGC, kmer), src/counting.jl:1-8 (2-bit alphabets)
        uint32_t n_gc = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) n_gc += __popcll((x[w] ^ (x[w] >> 1)) & 0x5555555555555555ull);
        y[0] = n_gc;
        return false;
    }
    if (op == 5) {  // LongSequence{A}(kmer).data, src/construction.jl:289-324:
        // move the unused bits to the bottom (_fill_shift!), then reverse the symbols of every word
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            uint64_t chunk = x[w];
            if (bu != 0) {
                chunk = x[w] << bu;
                if (w + 1 < NW) chunk |= x[w + 1] >> (64u - bu);
            }
            y[w] = reverse_symbols<BITS>(chunk);
        }
        return true;
    }
    if (op == 7 || op == 8) {
        // as_integer / from_integer (kmer.jl:305-326, :361-384): the value is the word tuple read as
        // one big-endian number; exported as u64 (NW == 1) or little-endian u128 (NW == 2).
        // from_integer keeps only the lowest K*bits bits (head word masked).
        if constexpr (NW == 2) {
            y[0] = x[1];
            y[1] = x[0];
            if (op == 8) y[0] &= mask;
        } else {
            if (op == 8) y[0] &= mask;
        }
        return true;
    }
    if (op == 0) {
        kmer_reverse<NW, BITS>(y, bu);
    } else if (op == 1) {
        kmer_complement<NW, BITS>(y, mask);
    } else {
        kmer_complement<NW, BITS>(y, mask);
        kmer_reverse<NW, BITS>(y, bu);  // reverse_complement = reverse(complement(x)), :32-34
    }
    if (op >= 3) {
        // lexicographic tuple compare, head first (kmer.jl:176-178)
        int c = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w)
            if (c == 0) c = x[w] < y[w] ? -1 : (x[w] > y[w] ? 1 : 0);
        if (op == 4) {
            y[0] = c <= 0 ? 1ull : 0ull;  // iscanonical: x <= rc (:41)
            return false;
        }
        if (c == -1) {  // canonical: ifelse(x < rc, x, rc) (:36-39)
#pragma unroll
            for (int w = 0; w < NW; ++w) y[w] = x[w];
        }
    }
    return true;
}

// VEC: 16-byte accesses (one-word kmers two per lane; two- and four-word kmers as ulonglong2);
// needs 16-byte aligned arrays.  One pass per workgroup (launch-shape lesson, r01_tuning.md).
template <int NW, int BITS, bool VEC>
__global__ __launch_bounds__(256) void transform_kernel(int op, const uint64_t *__restrict__ in, uint64_t n, int k,
                                                         uint64_t *__restrict__ out) {
    const uint64_t mask = head_mask(k, BITS);
    const uint32_t bu = (uint32_t)bits_unused(k, BITS);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC && NW == 1) {
        const uint64_t pairs = n / 2;
        for (uint64_t i = t0; i < pairs; i += stride) {
            ulonglong2 v = reinterpret_cast<const ulonglong2 *>(in)[i];
            uint64_t xa[1] = {v.x}, xb[1] = {v.y}, ya[1], yb[1];
            bool wide = transform_one<1, BITS>(op, xa, ya, mask, bu);
            transform_one<1, BITS>(op, xb, yb, mask, bu);
            (void)wide;  // one-word kmers: flags and kmers have the same 8-byte size
            reinterpret_cast<ulonglong2 *>(out)[i] = make_ulonglong2(ya[0], yb[0]);
        }
        if ((n & 1) && t0 == 0) {
            uint64_t xa[1] = {in[n - 1]}, ya[1];
            transform_one<1, BITS>(op, xa, ya, mask, bu);
            out[n - 1] = ya[0];
        }
        return;
    }
    for (uint64_t i = t0; i < n; i += stride) {
        uint64_t x[NW], y[NW];
        if constexpr (VEC && (NW == 2 || NW == 4)) {
#pragma unroll
            for (int w = 0; w < NW; w += 2) {
                ulonglong2 v = reinterpret_cast<const ulonglong2 *>(in + i * NW)[w / 2];
                x[w] = v.x;
                x[w + 1] = v.y;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) x[w] = in[i * NW + w];
        }
        if (!transform_one<NW, BITS>(op, x, y, mask, bu)) {
            out[i] = y[0];
            continue;
        }
        if constexpr (VEC && (NW == 2 || NW == 4)) {
#pragma unroll
            for (int w = 0; w < NW; w += 2)
                reinterpret_cast<ulonglong2 *>(out + i * NW)[w / 2] = make_ulonglong2(y[w], y[w + 1]);
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) out[i * NW + w] = y[w];
        }
    }
}

// ---- MinHash sketch maintenance, entirely on the device ------------------------------------------
// state[0] = number of values in best[], state[1] = threshold (values strictly below it are
// candidates), state[2] = overflow flag (a chunk produced more candidates than the buffer holds),
// state[3] = candidate counter.  One workgroup: merge best[] with the new candidates, keep the s
// smallest distinct values (ascending), publish the new threshold, reset the counter.
//
// Only the s smallest of the (typically 5-7 s) values matter, so the kernel first tries a cut: hashes
// are close to uniform below the old threshold, so a pivot at the (1.5 s + slack)/total quantile of
// [0, threshold) should leave about 1.5 s values; those are gathered, bitonic-sorted and deduplicated
// in LDS.  If they hold at least s distinct values they contain the answer.  Otherwise (skewed or
// heavily duplicated hashes, or too many values below the pivot) everything is sorted as before.
constexpr uint32_t SKETCH_LDS_VALUES = 16384;  // 128 KiB of dynamic LDS

// ascending bitonic sort of v[0..m) (m a power of two) by one 1024-thread workgroup.  Pair p of a
// step lives in elements [128 (p / 64), +128) whenever the partner distance j is <= 64, and pairs
// p, p + 1024, ... belong to the same wavefront, so consecutive steps with j <= 64 only exchange
// data inside one wavefront (whose LDS instructions execute in order): a workgroup barrier is
// needed only around the steps with j >= 128 -- 10 instead of 66 for m = 2048.
__device__ __forceinline__ void bitonic_sort_lds(uint64_t *v, uint32_t m, uint32_t t) {
    for (uint32_t k2 = 2; k2 <= m; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t p = t; p < (m >> 1); p += 1024) {  // one compare-exchange per pair
                const uint32_t i = ((p & ~(j - 1u)) << 1) | (p & (j - 1u)), l = i | j;
                const uint64_t a0 = v[i], a1 = v[l];
                const bool up = (i & k2) == 0;
                if ((a0 > a1) == up) { v[i] = a1; v[l] = a0; }
            }
            const uint32_t next_j = j > 1 ? j >> 1 : k2;  // the next phase starts at distance k2
            if (j > 64 || next_j > 64) block_sync();
            else __builtin_amdgcn_wave_barrier();
        }
    }
    block_sync();
}

__global__ __launch_bounds__(1024) void sketch_prune_kernel(uint64_t *__restrict__ best, uint64_t *__restrict__ state,
                                                             const uint64_t *__restrict__ cand, uint64_t cap, uint32_t s) {
    extern __shared__ uint64_t v[];            // SKETCH_LDS_VALUES values
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t sub_n;
    const uint32_t t = threadIdx.x;
    const uint32_t nb = (uint32_t)state[0];
    const uint64_t old_threshold = state[1];
    uint64_t cnt = state[3];
    if (cnt > cap) {
        if (t == 0) state[2] = 1;              // overflow: the host falls back to the feedback path
        cnt = cap;
    }
    const uint32_t total = nb + (uint32_t)cnt;
    auto value = [&](uint32_t i) { return i < nb ? best[i] : cand[i - nb]; };

    // sorts v[0..m) whose first `n` entries are real; returns the number of distinct values and, when
    // `commit`, writes the s smallest to best[] and publishes the threshold
    const uint32_t lane = t & 63u, wave = t >> 6;
    auto dedupe = [&](uint32_t m, uint32_t n, bool commit_if_enough, bool commit_always) -> uint32_t {
        bitonic_sort_lds(v, m, t);
        const uint32_t per = (m + 1023) / 1024;    // consecutive values per thread
        const uint32_t lo = t * per < n ? t * per : n, hi = lo + per < n ? lo + per : n;
        uint32_t mine = 0;
        for (uint32_t i = lo; i < hi; ++i) mine += (i == 0 || v[i] != v[i - 1]) ? 1u : 0u;
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t x = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += x;
        }
        block_sync();                           // wave_tot may still be read from an earlier call
        if (lane == 63) wave_tot[wave] = incl;
        block_sync();
        uint32_t before = 0, distinct = 0;
        for (uint32_t w = 0; w < 16; ++w) {
            if (w < wave) before += wave_tot[w];
            distinct += wave_tot[w];
        }
        if (commit_always || (commit_if_enough && distinct >= s)) {
            uint32_t pos = before + incl - mine;
            for (uint32_t i = lo; i < hi; ++i) {
                if (i == 0 || v[i] != v[i - 1]) {
                    if (pos < s) best[pos] = v[i];
                    if (pos + 1 == s) state[1] = v[i];  // the s-th smallest distinct value is the new threshold
                    ++pos;
                }
            }
            if (t == 0) {
                state[0] = distinct < s ? distinct : s;
                state[3] = 0;
                if (distinct < s) state[1] = ~0ull;     // fewer than s values so far: everything is still a candidate
            }
        }
        return distinct;
    };

    // ---- the cut ------------------------------------------------------------------------------
    // The first pivot assumes uniform hashes below the old threshold; if the count below it misses the
    // window [s, limit] (or duplicates leave fewer than s distinct values) the pivot is rescaled by the
    // observed density and the gather repeated, up to three times.
    const double target = 1.5 * (double)s + 8.0 * sqrt((double)s) + 32.0;
    uint32_t limit = 1;
    while ((double)limit < 1.25 * target) limit <<= 1;
    if ((double)total > 1.3 * (double)limit && limit <= SKETCH_LDS_VALUES) {
        double frac = target / (double)total;  // < 0.8
        for (int attempt = 0; attempt < 3; ++attempt) {
            const uint64_t pivot = frac >= 1.0 ? old_threshold : __umul64hi(old_threshold, (uint64_t)(frac * 18446744073709551616.0));
            if (t == 0) sub_n = 0;
            block_sync();
            for (uint32_t i = t; i < total; i += 1024) {
                const uint64_t x = value(i);
                if (x < pivot || frac >= 1.0) {
                    const uint32_t p = atomicAdd(&sub_n, 1u);
                    if (p < limit) v[p] = x;
                }
            }
            block_sync();
            const uint32_t c = sub_n;
            uint32_t distinct = 0;
            if (c >= s && c <= limit) {
                for (uint32_t i = c + t; i < limit; i += 1024) v[i] = ~0ull;
                block_sync();
                distinct = dedupe(limit, c, true, false);
                if (distinct >= s) return;  // uniform: every thread sees the same count
            }
            block_sync();
            if (frac >= 1.0) break;         // everything was below the pivot: nothing left to widen
            // too few (or too many duplicates): widen; too many: narrow -- by the observed density
            const double have = c > limit ? (double)c : (c >= s ? (double)distinct : (double)c);
            frac *= (c > limit ? 0.8 : 1.25) * target / (have > 1.0 ? have : 1.0);
            if (frac > 1.0) frac = 1.0;
        }
    }

    // ---- everything ---------------------------------------------------------------------------
    if (total > SKETCH_LDS_VALUES) {           // does not fit the LDS sort: the host falls back to the feedback path
        if (t == 0) state[2] = 1;
        return;
    }
    uint32_t m = 1;
    while (m < total) m <<= 1;                 // power of two >= total
    for (uint32_t i = t; i < m; i += 1024) v[i] = i < total ? value(i) : ~0ull;
    block_sync();
    dedupe(m, total, false, true);
}

// ---- one MinHash sketch per record of a batch (record_sketch_kernel.hpp) --------------------------
// One workgroup per record keeps the record's running bottom-s in LDS: hashes below the current threshold
// are appended behind it; when the next tile might not fit, the buffer is sorted, deduplicated and cut
// back to s values (the same merge the whole-sequence sketch does between rounds, per workgroup).
// Records with more than about 3 s hashes first try a provisional threshold at the (1.5 s + slack) / n
// quantile of the 64-bit range, which leaves about 1.8 s candidates -- one sweep and one small sort instead
// of sorting thousands of values to keep s; if fewer than s distinct values turn out to lie below it
// (skewed or heavily duplicated hashes) the record is swept again without it.
constexpr uint32_t SEG_VALUES = 8192;   // largest candidate buffer (64 KiB of dynamic LDS; 2048 or 4096 values are what calls use)
constexpr uint32_t SEG_UNROLL = 4;      // hashes per thread per tile

// ascending bitonic sort of v[0..m) by one 256-thread workgroup (same wave-local trick as above)
__device__ __forceinline__ void bitonic_sort_lds256(uint64_t *v, uint32_t m, uint32_t t) {
    for (uint32_t k2 = 2; k2 <= m; k2 <<= 1) {
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t p = t; p < (m >> 1); p += 256) {
                const uint32_t i = ((p & ~(j - 1u)) << 1) | (p & (j - 1u)), l = i | j;
                const uint64_t a0 = v[i], a1 = v[l];
                const bool up = (i & k2) == 0;
                if ((a0 > a1) == up) { v[i] = a1; v[l] = a0; }
            }
            const uint32_t next_j = j > 1 ? j >> 1 : k2;
            if (j > 64 || next_j > 64) block_sync();   // pairs p, p + 256, ... stay in one wavefront for j <= 64
            else __builtin_amdgcn_wave_barrier();
        }
    }
    block_sync();
}

// merge step of the per-record sketches: sorts v[0, total), keeps the s smallest distinct values in v[0, nb), returns nb
__device__ __forceinline__ uint32_t segment_merge(uint64_t *v, uint32_t total, uint32_t s, uint32_t t, uint32_t *wave_tot) {
    const uint32_t lane = t & 63u, wave = t >> 6;
    uint32_t m = 1;
    while (m < total) m <<= 1;
    for (uint32_t i = total + t; i < m; i += 256) v[i] = ~0ull;
    block_sync();
    bitonic_sort_lds256(v, m, t);
    // distinct values among the first `total`: thread t owns positions [a, b)
    const uint32_t per = (m + 255) / 256;
    const uint32_t a = t * per < total ? t * per : total, b = a + per < total ? a + per : total;
    uint64_t mine[SEG_VALUES / 256];
    uint32_t n_mine = 0;
    for (uint32_t i = a; i < b; ++i) {
        const uint64_t x = v[i];
        if (i == 0 || x != v[i - 1]) mine[n_mine++] = x;
    }
    uint32_t incl = n_mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t y = __shfl_up(incl, d, 64);
        if ((int)lane >= d) incl += y;
    }
    if (lane == 63) wave_tot[wave] = incl;
    block_sync();               // every thread has read its slice of v[]; wave totals visible
    uint32_t before = 0, distinct = 0;
    for (uint32_t w = 0; w < 4; ++w) {
        if (w < wave) before += wave_tot[w];
        distinct += wave_tot[w];
    }
    uint32_t pos = before + incl - n_mine;
    for (uint32_t i = 0; i < n_mine; ++i, ++pos)
        if (pos < s) v[pos] = mine[i];
    block_sync();
    return distinct < s ? distinct : s;
}

// ---- synthetic input (the CPU checker restates the same generator) ------
__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_word, uint64_t n_words, int bits,
                                                     uint32_t ambig, uint64_t *__restrict__ out) {
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        uint64_t w = first_word + i;
        if (bits == 2) {
            out[i] = synth_rand64(seed, w);
            continue;
        }
        uint64_t r = synth_rand64(seed, w >> 1) >> (32 * (w & 1));
        uint64_t word = 0;
        for (int j = 0; j < 16; ++j) {
            uint64_t nib = 1ull << ((r >> (2 * j)) & 3);
            if (ambig) {
                uint64_t b = w * 16 + (uint64_t)j;
                uint64_t u = (synth_rand64(seed ^ 0xA5A5A5A5A5A5A5A5ull, b >> 2) >> (16 * (b & 3))) & 0xffff;
                if (u < ambig) nib = 0xF;
            }
            word |= nib << (4 * j);
        }
        out[i] = word;
    }
}

}  // namespace kmers
