// scan_kernels.hpp -- the layout pass of kmers_batch (batch_api.hip) in ONE kernel: elements per record (FwKmers.jl:40-43,
// SpacedKmers.jl:38-42) and their exclusive scan, spans in, element offsets out.
//
// Rounds 1-4 ran four kernels for this (counts, per-segment sums, a scan of the sums, a rescan per segment: 82 us for 8 M
// records, the spans read once and the counts written once and read twice).  Round 5: a SEGMENT of 2048 x LAYOUT_CHUNKS records per
// workgroup, chained by a decoupled look-back -- a workgroup publishes its segment's AGGREGATE as soon as it
// has it, then sums the descriptors of the segments before it, nearest first, until it meets one that already holds an inclusive
// PREFIX; it never waits for a predecessor's prefix, only for aggregates, and every predecessor is a workgroup that is already
// running (segments are numbered by a ticket drawn at the start, not by blockIdx).  128 MB of spans read, 64 MB of offsets
// written, nothing else.  What it costs besides (tools/device_probes/layout_bench.hip, 8 M records): the tickets -- atomics on one address
// retire one per 7.5 ns, and a workgroup cannot load anything before it has its own: 3906 segments of 2048 paid 30 us for them --
// and the look-back's descriptor loads, which go past the L2 (agent scope): hence long segments, few descriptors, two words each.
//
// The descriptors need no clearing between calls: a descriptor is two 64-bit words, each carrying the call's EPOCH and the
// descriptor's state in its upper half and 32 bits of the value in its lower half, moved with relaxed agent-scope atomics; it
// counts only if both words carry the same tag of this call (a reader that catches an aggregate half way to becoming a prefix
// sees two tags and reads again).  The ticket counter only ever grows; the host passes the value it had before the launch.  The
// call's "a span reaches outside the pool" flag and the give-up flag of the look-back are written as the epoch itself.  So a
// call enqueues this kernel and nothing before it.
#pragma once
#include "ragged_kernels.hpp"

namespace kmers {

constexpr uint32_t LAYOUT_CHUNK = 2048;  // records per workgroup pass: 256 threads x 8
constexpr uint32_t LAYOUT_AGGREGATE = 1u, LAYOUT_PREFIX = 2u;
constexpr int LAYOUT_LOOKBACK = 4;     // descriptors per lane and look-back step
constexpr uint32_t LAYOUT_EPOCH_LIMIT = 1u << 30;  // (epoch << 2 | state fills the upper half of a descriptor word)
// header words of a call, in the context's device scratch (context.hpp): one copy brings them all to the host
constexpr int LAYOUT_WORD_TOTAL = 3, LAYOUT_WORD_BAD = 4, LAYOUT_WORD_ABORT = 5;
// Every spin is bounded (MI355X_MICROARCH.md, correctness boundaries; the same scheme as unambiguous_kernel.hpp): a look-back
// polls a missing aggregate at most LAYOUT_SPIN_LIMIT times -- seconds, where an aggregate is out microseconds after its ticket
// -- then raises the abort word, which the others check every LAYOUT_SPIN_CHECK polls; the host reports KMERS_E_HIP.
// The test build (kmers_jl_amd/build.py: -DKMERS_TEST_ABORT) never publishes segment 1 and gives up a thousand times sooner
// (tests/test_gpu_batch.py::test_batch_layout_gives_up_instead_of_hanging).
#ifdef KMERS_TEST_ABORT
constexpr uint32_t LAYOUT_SPIN_CHECK = 64, LAYOUT_SPIN_LIMIT = 1u << 12;
#else
constexpr uint32_t LAYOUT_SPIN_CHECK = 1024, LAYOUT_SPIN_LIMIT = 1u << 22;
#endif

struct LayoutArgs {
    const RaggedSpan *spans;
    uint64_t n;
    uint64_t pool_bases;
    unsigned long long *desc;    // [segments][2]: (tag | low half, tag | high half) of the aggregate, later of the prefix
    unsigned long long *ticket;  // grows by one per workgroup, never reset
    uint64_t ticket_base;        // its value before this launch
    uint64_t *offsets;           // [n + 1]; offsets[n] = the element count
    unsigned long long *header;  // LAYOUT_WORD_*
    uint32_t epoch;              // 1 .. LAYOUT_EPOCH_LIMIT - 1
    uint32_t k, step;
};

__device__ __forceinline__ unsigned long long layout_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void layout_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int C>  // chunks of LAYOUT_CHUNK records per segment
__global__ __launch_bounds__(256) void ragged_layout_kernel(const LayoutArgs a) {
    constexpr uint32_t PER = LAYOUT_CHUNK / 256;  // records per thread and chunk
    __shared__ uint32_t c[LAYOUT_CHUNK];
    __shared__ uint64_t o[LAYOUT_CHUNK];  // a chunk's offsets, so that they leave with the lanes side by side (a lane's own eight
                                          // consecutive offsets are 64 contiguous bytes per lane: the store shape that writes slowly)
    __shared__ uint64_t wave_tot[4];
    __shared__ uint64_t s_seg, s_base;
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    if (t == 0) s_seg = atomicAdd(a.ticket, 1ull) - a.ticket_base;
    block_sync();
    const uint64_t seg = s_seg;
    const uint64_t base = seg * (LAYOUT_CHUNK * C);
    const uint64_t n_seg = (a.n + LAYOUT_CHUNK * C - 1) / (LAYOUT_CHUNK * C);
    // ---- the segment's counts, in registers (lanes side by side: the loads are whole lines), and their sum
    uint32_t cnt[C][PER];
    bool bad = false;
    uint64_t mine_sum = 0;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint64_t i = base + (uint64_t)ch * LAYOUT_CHUNK + t + 256u * j;
            uint32_t v = 0;
            if (i < a.n) {
                const RaggedSpan sp = a.spans[i];
                if (sp.first_base > a.pool_bases || sp.n_bases > a.pool_bases - sp.first_base || sp.n_bases >= 0xFFFFFFFFull) bad = true;
                else if (sp.n_bases >= a.k) v = (uint32_t)((sp.n_bases - a.k) / a.step + 1u);
            }
            cnt[ch][j] = v;
            mine_sum += v;
        }
    }
    if (bad) a.header[LAYOUT_WORD_BAD] = a.epoch;
    for (int off = 32; off > 0; off >>= 1) mine_sum += __shfl_xor(mine_sum, off, 64);
    if (lane == 0) wave_tot[wave] = mine_sum;
    block_sync();
    if (wave == 0) {
        const uint64_t agg = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const uint32_t tag_a = a.epoch << 2 | LAYOUT_AGGREGATE, tag_p = a.epoch << 2 | LAYOUT_PREFIX;
        unsigned long long *const mine = a.desc + seg * 2;
        auto publish = [&](uint32_t tag, uint64_t v) {
            layout_store(mine, (uint64_t)tag << 32 | (uint64_t)(uint32_t)v);
            layout_store(mine + 1, (uint64_t)tag << 32 | (v >> 32));
        };
#ifdef KMERS_TEST_ABORT
        if (seg != 1)  // (nor its prefix below: segment 1 stays silent)
#endif
        if (lane == 0) publish(seg == 0 ? tag_p : tag_a, agg);
        uint64_t excl = 0;
        if (seg != 0) {
            long long pos = (long long)seg - 1;  // the nearest predecessor not yet accounted for
            uint64_t part = 0;
            bool found = false;
            while (!found) {
                // a step reads LAYOUT_LOOKBACK x 64 descriptors with all loads in flight together
                uint32_t st[LAYOUT_LOOKBACK];
                uint64_t val[LAYOUT_LOOKBACK];
                auto read = [&](int j) {
                    const long long idx = pos - (long long)(lane + 64u * (uint32_t)j);  // before segment 0: prefix 0
                    st[j] = LAYOUT_PREFIX;
                    val[j] = 0;
                    if (idx < 0) return;
                    const unsigned long long *d = a.desc + idx * 2;
                    const uint64_t w0 = layout_load(d), w1 = layout_load(d + 1);
                    const uint32_t tag = (uint32_t)(w0 >> 32);
                    if (tag == (uint32_t)(w1 >> 32) && (tag == tag_p || tag == tag_a)) {
                        st[j] = tag & 3u;
                        val[j] = (uint64_t)(uint32_t)w0 | w1 << 32;
                    } else {
                        st[j] = 0;
                    }
                };
#pragma unroll
                for (int j = 0; j < LAYOUT_LOOKBACK; ++j) read(j);
#pragma unroll
                for (int j = 0; j < LAYOUT_LOOKBACK; ++j) {
                    if (!found) {
                        uint32_t fp, spins = 0;
                        for (;;) {
                            const uint64_t bp = __ballot(st[j] == LAYOUT_PREFIX);
                            fp = bp ? (uint32_t)__builtin_ctzll(bp) : 64u;  // the nearest prefix of this window
                            const bool wait = st[j] == 0u && lane < fp;      // a nearer segment has not published yet
                            if (__ballot(wait) == 0) break;
                            __builtin_amdgcn_s_sleep(8);
                            if ((++spins % LAYOUT_SPIN_CHECK) == 0) {
                                const bool raised = layout_load(a.header + LAYOUT_WORD_ABORT) == (unsigned long long)a.epoch;
                                if (raised || spins >= LAYOUT_SPIN_LIMIT) {
                                    if (!raised && lane == 0) layout_store(a.header + LAYOUT_WORD_ABORT, (unsigned long long)a.epoch);
                                    fp = 0;  // give up: the host discards this call
                                    break;
                                }
                            }
                            if (wait) read(j);  // only the missing ones are read again
                        }
                        if (lane <= fp) part += val[j];  // aggregates up to and including the prefix
                        found = fp < 64u;
                    }
                }
                pos -= 64 * LAYOUT_LOOKBACK;
            }
            for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
            excl = part;
#ifdef KMERS_TEST_ABORT
            if (seg != 1)
#endif
            if (lane == 0) publish(tag_p, excl + agg);
        }
        if (lane == 0) {
            s_base = excl;
            if (seg == n_seg - 1) {
                a.offsets[a.n] = excl + agg;
                a.header[LAYOUT_WORD_TOTAL] = excl + agg;
            }
        }
    }
    block_sync();
    // ---- chunk by chunk: the counts change hands through LDS (thread t owns PER consecutive ones), local exclusive prefix, a scan
    //      over the thread totals, and the offsets leave through LDS again
    uint64_t carry = s_base;
#pragma unroll
    for (int ch = 0; ch < C; ++ch) {
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) c[t + 256u * j] = cnt[ch][j];
        block_sync();
        uint64_t local[PER];  // 64-bit: per-record counts of up to 2^32 - 1
        uint64_t sum = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            local[j] = sum;
            sum += c[t * PER + j];
        }
        uint64_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint64_t v = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        block_sync();
        uint64_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
        carry += wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        const uint64_t excl_t = before + incl - sum;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) o[t * PER + j] = excl_t + local[j];
        block_sync();
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint64_t i = base + (uint64_t)ch * LAYOUT_CHUNK + t + 256u * j;
            if (i < a.n) a.offsets[i] = o[t + 256u * j];
        }
        if (ch + 1 < C) block_sync();  // (the next chunk reuses c, o and the wave totals)
    }
}

// Chunks per segment (tools/device_probes/layout_bench.hip; 1 / 2 / 4 / 8 chunks): 1 M records 21 / 18 / 22 / 34 us, 8 M 90 / 65 / 63 / 88 us,
// 40 M 346 / 302 / 354 / 569 us -- short segments pay for tickets and descriptors, long ones serialise their own phases.
constexpr int LAYOUT_CHUNKS = 2;

}  // namespace kmers
