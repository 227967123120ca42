// stream_kernel.hpp -- the tile kernel behind kmers_fw / kmers_canonical / kmers_spaced /
// kmers_reduce_xor (reference: src/iterators/FwKmers.jl:57-115, CanonicalKmers.jl:54-144,
// :220-225, SpacedKmers.jl:83-139, src/kmer.jl:255-261).
//
// One workgroup (256 threads = 4 wavefronts of 64) owns a tile of `tile_kmers` consecutive
// kmer indices.  Phase 1 reads the source words the tile touches once (coalesced 8-byte
// loads), turns 4-bit one-hot symbols into 2-bit codes (validating count_ones == 1) and
// stages the 2-bit little-endian stream -- the tile plus its (K-1)-base overlap into the
// next tile -- in LDS.  Phase 2 cuts every window out of that stream with a 64-bit funnel
// shift, forms forward / reverse-complement / canonical / fx_hash in registers and writes
// 16 bytes per lane per store instruction, lanes consecutive, so every wave store covers
// whole 128-byte lines.  HBM-bound on the output: see DESIGN.md for bytes per kmer.
#pragma once
#include "device_bits.hpp"

namespace kmers {

#ifndef KMERS_BLOCK
#define KMERS_BLOCK 256
#endif
constexpr int BLOCK = KMERS_BLOCK;             // threads per workgroup (multiple of 64)
constexpr int WAVES = BLOCK / 64;
constexpr int MAX_TILE_BASES = 16384;          // bases of 2-bit stream a tile may span (excl. overlap)
constexpr int LDS_QWORDS = MAX_TILE_BASES / 32 + 16;

enum Mode { MODE_FW = 0, MODE_CANON = 1, MODE_XOR = 2 };

struct StreamArgs {
    const uint64_t *src;     // LongSequence.data in HBM
    uint64_t first_bit;      // bit offset of symbol 1 inside src
    uint64_t n_bases;        // symbols in the view
    uint64_t n_kmers;        // elements to produce
    uint64_t inspect_end;    // symbols [0, inspect_end) are validated (see kernel)
    uint64_t *out_a;         // FW: forward kmers   CANON: canonical kmers (nullable)  XOR: accumulator
    uint64_t *out_b;         // FW: reverse complements (nullable)   CANON: hashes (nullable)
    uint64_t seed;           // fx_hash seed
    unsigned long long *err_slot;  // atomicMin of the first offending symbol (0-based)
    uint64_t n_tiles;
    uint32_t k;
    uint32_t stride;
    uint32_t tile_kmers;
    uint32_t xor_canonical;  // MODE_XOR: 1 = canonical kmers, 0 = forward kmers
};

// First inspected ambiguous symbol of one 4-bit source word -> err_slot (rare path, kept
// inline and call-free so the kernel needs no stack).  Inspected set = what the reference's
// iterate() would have looked at before stopping: every symbol below inspect_end, except
// (stride >= K) the gaps between kmers (src/iterators/SpacedKmers.jl:133-134).
template <bool STRIDE1>
__device__ __forceinline__ void report_ambiguous(unsigned long long *err_slot, uint64_t first_bit,
                                                 uint64_t inspect_end, uint32_t stride, uint32_t k,
                                                 uint64_t word_index, uint64_t bad) {
    uint64_t f = (bad | (bad >> 1) | (bad >> 2)) & 0x1111111111111111ull;  // bit 4j = symbol j is ambiguous
    // symbol index of nibble 0 of this word; negative inside the first word of an offset view
    // (both terms are multiples of 4, so the division is exact)
    const long long base0 = ((long long)(word_index * 64) - (long long)first_bit) / 4;
    if (base0 < 0) f &= ~0ull << (uint32_t)(-base0 * 4);
    const long long room = (long long)inspect_end - base0;  // symbols of this word below inspect_end
    if (room <= 0) return;
    if (room < 16) f &= (1ull << (uint32_t)(room * 4)) - 1ull;
    if constexpr (!STRIDE1) {  // stride 1 has no gaps
        if (stride >= k) {
#pragma unroll 1
            for (uint32_t j = 0; j < 16; ++j)
                if (((f >> (4 * j)) & 1ull) && ((uint64_t)(base0 + j) % stride) >= k) f &= ~(1ull << (4 * j));
        }
    }
    if (f) atomicMin(err_slot, (unsigned long long)(base0 + (long long)(__ffsll((long long)f) - 1) / 4));
}

// forward / reverse-complement kmers (N words, head first) of the window whose first base
// sits at stream bit `bit` of the LDS stream.
template <int N>
__device__ __forceinline__ void window(const uint64_t *lds, uint32_t bit, uint32_t k, uint64_t mask,
                                       uint64_t (&fw)[N], uint64_t (&rc)[N]) {
    uint32_t q = bit >> 6, s = bit & 63u;
    if constexpr (N == 1) {
        uint64_t W = funnel64(lds[q], lds[q + 1], s) & mask;
        rc[0] = ~W & mask;
        fw[0] = rev2(W) >> (64u - 2u * k);
    } else {
        static_assert(N == 2, "window: N must be 1 or 2");
        uint64_t q0 = lds[q], q1 = lds[q + 1], q2 = lds[q + 2];
        uint64_t Wlo = funnel64(q0, q1, s);
        uint64_t Whi = funnel64(q1, q2, s) & mask;  // mask covers the 2K-64 bits of the head word
        rc[0] = ~Whi & mask;
        rc[1] = ~Wlo;
        // 128-bit symbol reversal then right shift by 128-2K (0..62)
        uint64_t hi = rev2(Wlo), lo = rev2(Whi);
        uint32_t sh = 128u - 2u * k;
        fw[1] = (lo >> sh) | ((hi << 1) << (63u - sh));
        fw[0] = hi >> sh;
    }
}

template <int N>
__device__ __forceinline__ bool kmer_less(const uint64_t (&x)[N], const uint64_t (&y)[N]) {
    if constexpr (N == 1) return x[0] < y[0];
    else return x[0] < y[0] || (x[0] == y[0] && x[1] < y[1]);  // cmp(x.data, y.data) == -1, kmer.jl:176-178
}

template <int N>
__device__ __forceinline__ uint64_t fx_hash(const uint64_t (&x)[N], uint64_t seed) {
    uint64_t h = seed;
#pragma unroll
    for (int i = 0; i < N; ++i) h = fx_step(h, x[i]);
    return h;
}

template <int SRC_BITS, int N, int MODE, bool STRIDE1>
__global__ __launch_bounds__(BLOCK) void stream_kernel(const StreamArgs a) {
    __shared__ uint64_t lds[LDS_QWORDS];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint32_t J = STRIDE1 ? 1u : a.stride;
    // N == 1: mask of the single word; N == 2: mask of the head word (2K-64 bits)
    const uint64_t mask = head_mask((int)k, 2);
    constexpr uint32_t KPL = (STRIDE1 && N == 1) ? 2u : 1u;  // kmers per lane per pass -> 16 B stores
    uint64_t xacc = 0;

    for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint64_t m0 = tile * a.tile_kmers;
        const uint64_t left = a.n_kmers - m0;
        const uint32_t mt = left < a.tile_kmers ? (uint32_t)left : a.tile_kmers;
        const uint64_t bit0 = a.first_bit + m0 * J * SRC_BITS;
        const uint64_t w0 = bit0 >> 6;
        const uint32_t b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(mt - 1) * J + k) * SRC_BITS;
        const uint32_t nw = (uint32_t)(((end_bit + 63) >> 6) - w0);

        __syncthreads();  // previous tile's readers are done with the LDS stream
        // ---- phase 1: source words -> 2-bit stream in LDS --------------------------------
        for (uint32_t wi = tid; wi < nw; wi += BLOCK) {
            uint64_t x = a.src[w0 + wi];
            if constexpr (SRC_BITS == 4) {
                uint64_t bad;
                uint32_t c = pack_4to2(x, bad);
                reinterpret_cast<uint32_t *>(lds)[wi] = c;
                if (bad) report_ambiguous<STRIDE1>(a.err_slot, a.first_bit, a.inspect_end, a.stride, k, w0 + wi, bad);
            } else {
                lds[wi] = x;
            }
        }
        __syncthreads();

        // ---- phase 2: windows -> kmers ---------------------------------------------------
        for (uint32_t r = tid * KPL; r < mt; r += BLOCK * KPL) {
            const uint64_t g = m0 + r;
            uint64_t fw[KPL][N], rc[KPL][N];
            window<N>(lds, 2u * (r * J + b0), k, mask, fw[0], rc[0]);
            if constexpr (KPL == 2) {
                // next window: one symbol further.  rc shifts right, fw shifts left
                // (the reference's own rolling step, CanonicalKmers.jl:102-103).
                uint32_t bit = 2u * (r + b0 + k);
                uint64_t code = (lds[bit >> 6] >> (bit & 63u)) & 3u;
                fw[1][0] = ((fw[0][0] << 2) | code) & mask;
                rc[1][0] = (rc[0][0] >> 2) | ((code ^ 3u) << (2u * k - 2u));
            }
            const bool both = (KPL == 2) && (r + 1 < mt);

            if constexpr (MODE == MODE_FW) {
                if constexpr (KPL == 2) {
                    if (both) {
                        *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(fw[0][0], fw[1][0]);
                        if (a.out_b) *reinterpret_cast<ulonglong2 *>(a.out_b + g) = make_ulonglong2(rc[0][0], rc[1][0]);
                    } else {
                        a.out_a[g] = fw[0][0];
                        if (a.out_b) a.out_b[g] = rc[0][0];
                    }
                } else if constexpr (N == 1) {
                    a.out_a[g] = fw[0][0];
                    if (a.out_b) a.out_b[g] = rc[0][0];
                } else {
                    *reinterpret_cast<ulonglong2 *>(a.out_a + 2 * g) = make_ulonglong2(fw[0][0], fw[0][1]);
                    if (a.out_b) *reinterpret_cast<ulonglong2 *>(a.out_b + 2 * g) = make_ulonglong2(rc[0][0], rc[0][1]);
                }
            } else {
                // canonical: fw < rv ? fw : rv (CanonicalKmers.jl:224)
                uint64_t c[KPL][N];
#pragma unroll
                for (uint32_t e = 0; e < KPL; ++e) {
                    bool lt = kmer_less<N>(fw[e], rc[e]);
#pragma unroll
                    for (int w = 0; w < N; ++w) c[e][w] = lt ? fw[e][w] : rc[e][w];
                }
                if constexpr (MODE == MODE_XOR) {
                    const bool can = a.xor_canonical != 0;
                    xacc ^= can ? c[0][0] : fw[0][0];
                    if (both) xacc ^= can ? c[1][0] : fw[1][0];
                } else if constexpr (KPL == 2) {
                    if (both) {
                        if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + g) = make_ulonglong2(c[0][0], c[1][0]);
                        if (a.out_b)
                            *reinterpret_cast<ulonglong2 *>(a.out_b + g) =
                                make_ulonglong2(fx_hash<N>(c[0], a.seed), fx_hash<N>(c[1], a.seed));
                    } else {
                        if (a.out_a) a.out_a[g] = c[0][0];
                        if (a.out_b) a.out_b[g] = fx_hash<N>(c[0], a.seed);
                    }
                } else if constexpr (N == 1) {
                    if (a.out_a) a.out_a[g] = c[0][0];
                    if (a.out_b) a.out_b[g] = fx_hash<N>(c[0], a.seed);
                } else {
                    if (a.out_a) *reinterpret_cast<ulonglong2 *>(a.out_a + 2 * g) = make_ulonglong2(c[0][0], c[0][1]);
                    if (a.out_b) a.out_b[g] = fx_hash<N>(c[0], a.seed);
                }
            }
        }
    }

    if constexpr (MODE == MODE_XOR) {
        // wavefront XOR-reduce (64 lanes), then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if ((tid & 63u) == 0) atomicXor(reinterpret_cast<unsigned long long *>(a.out_a), (unsigned long long)xacc);
    }
}

// Direct (gather) kernel for large strides, where a tile would stage mostly unused bases:
// one lane per kmer, symbols fetched one by one exactly like unsafe_extract
// (src/construction_utils.jl:41-69).  Edge path, not bandwidth critical.
template <int SRC_BITS, int N>
__global__ __launch_bounds__(BLOCK) void gather_kernel(const StreamArgs a) {
    uint64_t g = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= a.n_kmers) return;
    uint64_t base = g * a.stride;
    uint64_t d[N];
#pragma unroll
    for (int w = 0; w < N; ++w) d[w] = 0;
    for (uint32_t t = 0; t < a.k; ++t) {
        uint64_t bit = a.first_bit + (base + t) * SRC_BITS;
        uint64_t enc = (a.src[bit >> 6] >> (bit & 63u)) & ((1u << SRC_BITS) - 1u);
        uint64_t code = enc;
        if constexpr (SRC_BITS == 4) {
            if (__popcll(enc) != 1) {
                atomicMin(a.err_slot, (unsigned long long)(base + t));
                return;
            }
            code = (uint64_t)(__ffsll((long long)enc) - 1);
        }
        if constexpr (N == 2) d[0] = (d[0] << 2) | (d[1] >> 62);
        d[N - 1] = (d[N - 1] << 2) | code;
    }
#pragma unroll
    for (int w = 0; w < N; ++w) a.out_a[g * N + w] = d[w];
}

}  // namespace kmers
