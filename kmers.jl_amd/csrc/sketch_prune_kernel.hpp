// sketch_prune_kernel.hpp -- the one-workgroup merge of the whole-sequence MinHash sketch (consumers_api.hip only).
#pragma once
#include "device_bits.hpp"

namespace kmers {

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

}  // namespace kmers
