// record_sketch_kernel.hpp -- one MinHash sketch per record of a batch, fused: MinHash.jl's
// `sketch(fx_hash, CanonicalKmers{A,K}(record), s)` for every record of a pool (docs/src/minhash.md:31-41)
// without materialising the hashes.  One workgroup per record walks the record in tiles of 1024 windows:
// the tile's stretch of the (recoded) stream is staged in LDS, every lane derives four consecutive
// canonical kmers (first window cut out of the staged words, the others by the reference's rolling step,
// CanonicalKmers.jl:131-144) and their fx_hash (kmer.jl:255-261), and the values below the record's running
// threshold are appended to the candidate buffer described in segment_sort.hpp (segment_merge).
// Nothing but the pool is read and nothing but the sketches is written: 0.25-0.5 B per base instead of the
// 16 B per kmer of hashes written and read back.
#pragma once
#include "segment_sort.hpp"
#include "ragged_kernels.hpp"

namespace kmers {

// windows per tile = 256 * RUN; RUN = 4 (short records: the candidate buffer needs room for the sketch and one tile)
// or 8 (long records: half the per-run set-up and half the barriers per window)
constexpr uint32_t RS_STAGE = 144;               // stream words of a tile: (2048 + 127 symbols) * 4 bits / 64 + slack
constexpr uint32_t RS_FSTAGE = 40;               // flag words of a tile

struct RecordSketchArgs {
    const uint64_t *stream;      // DST-bit symbol stream (the pool itself for Copyable pools)
    const uint64_t *flags;       // one bit per stream symbol, NULL when nothing can fail
    const uint64_t *any_flag;    // set by the recode pass if it flagged any symbol
    uint64_t stream_origin;      // stream symbol index of pool symbol 0
    const RaggedSpan *spans;
    uint64_t pool_bases;
    uint64_t seed;
    uint64_t *out;               // [n_records * s]
    uint64_t *counts;            // [n_records]
    unsigned long long *err_slot;  // atomicMin of (record << 32 | window) of a window over a flagged symbol (strict mode)
    uint64_t *bad;               // becomes non-zero if a span reaches outside the pool (or holds 2^32 symbols or more)
    uint32_t k, s, skip, cap;
    uint32_t n_words;            // N == 0 (kmers of more than four words): the run-time width
    uint64_t rec_base;           // workgroup b of the launch takes record rec_base + b (a batch launched in pieces, batch_api.hip)
};

// N == 0: kmers of any width (a.n_words); nothing is staged, every window is read from the stream in HBM (an edge path).
template <int DST, int N, int RUN>
__global__ __launch_bounds__(256) void record_sketch_kernel(const RecordSketchArgs a) {
    constexpr uint32_t RS_TILE = 256u * RUN;
    extern __shared__ uint64_t v[];            // cap candidate values, then the staging area
    __shared__ uint32_t fill;
    __shared__ uint32_t wave_tot[4];
    uint64_t *src_t = v + a.cap;
    uint64_t *flg_t = src_t + RS_STAGE;
    const uint32_t t = threadIdx.x;
    const uint64_t r = a.rec_base + blockIdx.x;
    const uint32_t k = a.k, s = a.s;
    const RaggedSpan sp = a.spans[r];
    if (sp.first_base > a.pool_bases || sp.n_bases > a.pool_bases - sp.first_base || sp.n_bases >= 0xFFFFFFFFull) {
        if (t == 0) {
            a.bad[0] = 1;
            a.counts[r] = 0;
        }
        return;
    }
    const uint64_t n = sp.n_bases < k ? 0 : sp.n_bases - k + 1;   // windows of the record (FwKmers.jl:40-43)
    const uint64_t p0 = sp.first_base + a.stream_origin;          // stream symbol of the record's first
    const uint64_t *flags = (a.flags && *a.any_flag) ? a.flags : nullptr;
    const uint64_t mask = head_mask((int)k, DST);
    uint32_t nb = 0;
    for (int pass = 0; pass < 2; ++pass) {     // pass 0: provisional threshold (segment_sort.hpp); pass 1: without it
        const double frac = n ? (1.5 * (double)s + 8.0 * sqrt((double)s) + 32.0) / (double)n : 1.0;
        const bool provisional = pass == 0 && frac < 0.5;
        uint64_t threshold = ~0ull;
        if (provisional) threshold = (uint64_t)(frac * 18446744073709551616.0);
        else if (pass == 0) pass = 1;
        nb = 0;
        if (t == 0) fill = 0;
        block_sync();
        for (uint64_t base = 0; base < n || base == 0; base += RS_TILE) {
            if (base < n) {
                const uint32_t nel = n - base < (uint64_t)RS_TILE ? (uint32_t)(n - base) : RS_TILE;
                const uint64_t ps = p0 + base;                    // first symbol of the tile
                const uint64_t q0 = (ps * (uint64_t)DST) >> 6;
                const uint32_t nwords = (uint32_t)((((ps + nel + k - 1u) * (uint64_t)DST + 63u) >> 6) - q0);
                const uint64_t f0 = ps >> 6;
                if constexpr (N != 0) {
                    for (uint32_t i = t; i < nwords; i += 256u) src_t[i] = a.stream[q0 + i];
                    if (flags) {
                        const uint32_t nf = (uint32_t)(((ps + nel + k - 1u + 63u) >> 6) - f0);
                        for (uint32_t i = t; i < nf; i += 256u) flg_t[i] = flags[f0 + i];
                    }
                }
                block_sync();  // (also: every lane has read `fill` of the previous tile before anyone appends again)
                const uint32_t e = (uint32_t)RUN * t;                // this lane's first window of the tile
                if (e < nel) {
                    const uint32_t cnt = nel - e < (uint32_t)RUN ? nel - e : (uint32_t)RUN;
                    uint64_t hv[(uint32_t)RUN];
                    bool keep[(uint32_t)RUN];
                    const uint64_t p = ps + e;
                    const uint64_t bit = p * (uint64_t)DST;
                    const uint32_t rel = (uint32_t)((bit >> 6) - q0);
                    const uint32_t sh = (uint32_t)(bit & 63u);
                    if constexpr (N == 0) {
#pragma unroll 1
                        for (uint32_t j = 0; j < (uint32_t)RUN; ++j) {
                            keep[j] = false;
                            hv[j] = 0;
                            if (j < cnt) {
                                const uint64_t pj = p + j;
                                bool flagged = false;
                                if (flags) {
                                    const uint64_t fq = pj >> 6;
                                    flagged = any_flag_in([&](uint32_t i) { return flags[fq + i]; }, (uint32_t)(pj & 63u), k);
                                }
                                if (flagged && !a.skip) atomicMin(a.err_slot, (unsigned long long)((r << 32) | (base + e + j)));
                                auto load = [&](uint64_t q) -> uint64_t { return a.stream[q]; };
                                const bool take_fw = wide_forward_is_canonical_from_stream<DST>(load, pj, k, a.n_words);  // CanonicalKmers.jl:220-225
                                uint64_t h = a.seed;
                                for (uint32_t w = 0; w < a.n_words; ++w) h = fx_step(h, wide_word_from_stream<DST>(load, pj, k, a.n_words, w, !take_fw));
                                hv[j] = h;
                                keep[j] = !flagged;
                            }
                        }
                    } else if constexpr (N == 1) {
                        const uint32_t span = k + cnt - 1u;       // symbols the run reads
                        uint64_t fbits = 0;                       // flagged symbols of the run
                        if (flags) {
                            const uint32_t fr = (uint32_t)((p >> 6) - f0), fs = (uint32_t)(p & 63u);
                            uint64_t f = flg_t[fr] >> fs;
                            if (fs + span > 64u) f |= (flg_t[fr + 1u] << 1) << (63u - fs);
                            fbits = f & ((1ull << span) - 1ull);  // span <= 32 + 7
                        }
                        const uint64_t kbits = k >= 64u ? ~0ull : (1ull << k) - 1ull;
                        const uint32_t need = (sh + (uint32_t)DST * span + 63u) >> 6;  // 1..3 stream words
                        const uint64_t l0 = src_t[rel], l1 = need > 1u ? src_t[rel + 1u] : 0, l2 = need > 2u ? src_t[rel + 2u] : 0;
                        const uint64_t W0 = funnel64(l0, l1, sh), W1 = funnel64(l1, l2, sh);
                        uint64_t fw[1], rc[1];
                        fw[0] = rev_symbols<DST>(W0 & mask) >> (64u - (uint32_t)DST * k);
                        rc[0] = comp_symbols<DST>(W0 & mask);
                        if constexpr (DST == 2) rc[0] &= mask;
                        const uint32_t S = (uint32_t)((uint32_t)DST * k == 64u ? W1 : funnel64(W0, W1, (uint32_t)DST * k));
                        const uint32_t top = (uint32_t)DST * (k - 1u);
#pragma unroll
                        for (uint32_t j = 0; j < (uint32_t)RUN; ++j) {
                            keep[j] = false;
                            hv[j] = 0;
                            if (j < cnt) {
                                if (j > 0) {
                                    const uint64_t sym = (S >> ((uint32_t)DST * (j - 1u))) & ((1u << DST) - 1u);
                                    uint64_t csym;
                                    if constexpr (DST == 2) csym = sym ^ 3u;
                                    else csym = ((sym & 1u) << 3) | ((sym & 2u) << 1) | ((sym & 4u) >> 1) | ((sym & 8u) >> 3);
                                    fw[0] = ((fw[0] << DST) | sym) & mask;
                                    rc[0] = (rc[0] >> DST) | (csym << top);
                                }
                                const bool flagged = ((fbits >> j) & kbits) != 0;
                                if (flagged && !a.skip) atomicMin(a.err_slot, (unsigned long long)((r << 32) | (base + e + j)));
                                uint64_t x[1] = {kmer_less<1>(fw, rc) ? fw[0] : rc[0]};   // CanonicalKmers.jl:220-225
                                hv[j] = fx_hash<1>(x, a.seed);
                                keep[j] = !flagged;
                            }
                        }
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < (uint32_t)RUN; ++j) {
                            keep[j] = false;
                            hv[j] = 0;
                            if (j < cnt) {
                                const uint64_t pj = p + j;
                                const uint64_t bj = pj * (uint64_t)DST;
                                const uint32_t rj = (uint32_t)((bj >> 6) - q0);
                                bool flagged = false;
                                if (flags) {
                                    const uint32_t fr = (uint32_t)((pj >> 6) - f0);
                                    flagged = any_flag_in([&](uint32_t i) { return flg_t[fr + i]; }, (uint32_t)(pj & 63u), k);
                                }
                                if (flagged && !a.skip) atomicMin(a.err_slot, (unsigned long long)((r << 32) | (base + e + j)));
                                uint64_t fw[N], rc[N], x[N];
                                window_words<N, DST>([&](uint32_t i) { return src_t[rj + i]; }, (uint32_t)(bj & 63u), k, mask, fw, rc);
                                const bool lt = kmer_less<N>(fw, rc);
#pragma unroll
                                for (int w = 0; w < N; ++w) x[w] = lt ? fw[w] : rc[w];
                                hv[j] = fx_hash<N>(x, a.seed);
                                keep[j] = !flagged;
                            }
                        }
                    }
                    (void)rel;
                    (void)sh;
#pragma unroll
                    for (uint32_t j = 0; j < (uint32_t)RUN; ++j)  // room for a whole tile is guaranteed by the merge condition below
                        if (keep[j] && (hv[j] < threshold || (!provisional && nb < s))) v[nb + atomicAdd(&fill, 1u)] = hv[j];
                }
            }
            block_sync();
            const bool last = base + RS_TILE >= n;
            const uint32_t total = nb + fill;  // uniform
            if (last || total + RS_TILE > a.cap) {
                nb = segment_merge(v, total, s, t, wave_tot);
                if (nb == s && v[s - 1] < threshold) threshold = v[s - 1];
                if (t == 0) fill = 0;
                block_sync();
            }
            if (last) break;
        }
        if (!provisional || nb == s) break;
    }
    for (uint32_t i = t; i < nb; i += 256) a.out[r * (uint64_t)s + i] = v[i];
    if (t == 0) a.counts[r] = nb;
}

}  // namespace kmers
