// scan_kernels.hpp -- exclusive scan of per-record element counts: the layout pass of kmers_batch (batch_api.hip).
#pragma once
#include "device_bits.hpp"

namespace kmers {

// ---- exclusive scan of n 32-bit counts into 64-bit offsets (offsets[n] = total): the layout pass of kmers_batch ----
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
    __shared__ uint64_t o[SCAN_SEG];  // the segment's offsets, so that they leave with the lanes side by side (a lane's own eight
                                      // consecutive offsets are 64 contiguous bytes per lane: the store shape that writes slowly)
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
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) o[t * (SCAN_SEG / 256) + j] = excl + local[j];
    block_sync();
#pragma unroll
    for (uint32_t j = 0; j < SCAN_SEG / 256; ++j) {
        const uint64_t i = base + t + 256u * j;
        if (i < n) offsets[i] = o[t + 256u * j];
    }
    if (blockIdx.x == 0 && t == 0) offsets[n] = seg_sums[n_seg];
}

}  // namespace kmers
