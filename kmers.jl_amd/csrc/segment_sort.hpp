// segment_sort.hpp -- the per-workgroup candidate buffer of record_sketch_kernel.hpp: sort, deduplicate, cut back to s.
#pragma once
#include "device_bits.hpp"

namespace kmers {

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

}  // namespace kmers
