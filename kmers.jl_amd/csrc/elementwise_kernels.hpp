// elementwise_kernels.hpp -- element-wise kernels over arrays of kmers and the synthetic input generator (elementwise_api.hip
// only: the non-template kernels here must live in exactly one translation unit).  Reference: src/kmer.jl:255-261 (fx_hash),
// src/transformations.jl:1-41 (reverse / complement / reverse_complement / canonical / iscanonical).
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

// Kmers of more than four words (Kmer{A,K,N} has no bound on N, src/kmer.jl:97-111): the width is a run-time argument, a
// word of the result is computed from the one or two input words it depends on, nothing is held in a register array.
// One lane per kmer, 8-byte accesses `nw` words apart: an edge path (2 KiB kmers are not a throughput workload).
template <int BITS>
__global__ __launch_bounds__(256) void transform_kernel_any(int op, const uint64_t *__restrict__ in, uint64_t n, int k, int nw,
                                                             uint64_t *__restrict__ out) {
    const uint64_t mask = head_mask(k, BITS);
    const uint32_t bu = (uint32_t)bits_unused(k, BITS);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t *x = in + i * (uint64_t)nw;
        // word w of complement(x): transformations.jl:14-25
        auto comp = [&](int w) -> uint64_t {
            const uint64_t c = complement_word<BITS>(x[w]);
            return (BITS == 2 && w == 0) ? (c & mask) : c;
        };
        // word w of reverse(v), v given word by word: reversebits of every word, tuple reversed, right shift by bits_unused
        // (transformations.jl:1-10)
        auto rev = [&](auto &&v, int w) -> uint64_t {
            const uint64_t t = reverse_symbols<BITS>(v(nw - 1 - w));
            const uint64_t above = w > 0 ? ((reverse_symbols<BITS>(v(nw - w)) << 1) << (63u - bu)) : 0ull;
            return (t >> bu) | above;
        };
        auto plain = [&](int w) -> uint64_t { return x[w]; };
        if (op == 6) {  // count(isGC, kmer), src/counting.jl:1-8
            uint64_t n_gc = 0;
            for (int w = 0; w < nw; ++w) n_gc += (uint64_t)__popcll((x[w] ^ (x[w] >> 1)) & 0x5555555555555555ull);
            out[i] = n_gc;
            continue;
        }
        if (op == 3 || op == 4) {  // canonical / iscanonical: lexicographic compare with the reverse complement, head first
            int c = 0;
            for (int w = 0; w < nw && c == 0; ++w) {
                const uint64_t r = rev(comp, w);
                c = x[w] < r ? -1 : (x[w] > r ? 1 : 0);
            }
            if (op == 4) {
                out[i] = c <= 0 ? 1ull : 0ull;
                continue;
            }
            uint64_t *y = out + i * (uint64_t)nw;
            for (int w = 0; w < nw; ++w) y[w] = c == -1 ? x[w] : rev(comp, w);
            continue;
        }
        uint64_t *y = out + i * (uint64_t)nw;
        for (int w = 0; w < nw; ++w) {
            uint64_t v;
            if (op == 0) v = rev(plain, w);
            else if (op == 1) v = comp(w);
            else if (op == 2) v = rev(comp, w);
            else {  // 5: LongSequence{A}(kmer).data, src/construction.jl:289-324
                uint64_t chunk = x[w];
                if (bu != 0) {
                    chunk = x[w] << bu;
                    if (w + 1 < nw) chunk |= x[w + 1] >> (64u - bu);
                }
                v = reverse_symbols<BITS>(chunk);
            }
            y[w] = v;
        }
    }
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
