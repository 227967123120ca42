// run_kernel.hpp -- fused consumers of one- and two-word 2-bit kmers (K <= 64) that keep nothing per kmer:
//   RMODE_XOR    the reference's own benchmark consumer (test/benchmark.jl:9-15): XOR of the
//                canonical (or forward) kmers' data words;
//   RMODE_SKETCH MinHash candidates: fx_hash(canonical kmer) below the running threshold
//                (docs/src/minhash.md:31-35).
// These are issue-bound (0.25-0.5 B of source per kmer), so the kernel spends as few issue CYCLES per kmer as it can
// (profiles/r04_valu_rates.txt: a SIMD of gfx950 takes a simple two-operand integer instruction every 2.24 cycles and every
// other vector instruction every 4.1).  A lane takes a RUN of 32 consecutive kmers from the N + 1 window words of the LDS
// stream it needs anyway (K + 31 <= 64 N + 31 symbols).  One-word kmers of a full run are CUT as windows of two 128-bit
// streams, two v_alignbit each (see the tile loop); two-word kmers and the last, short run of a sequence are built once
// (symbol reversal, complement) and then ROLLED one symbol at a time -- shift_encoding / shift_first_encoding, exactly the step
// of CanonicalKmers.jl:131-144.  The next tile's source words travel from HBM while the current tile is consumed (register
// prefetch).  profiles/r04_fused.md: what the kernel costs per phase, and the structural experiments that did not help.
#pragma once
#include "stream_kernel.hpp"

namespace kmers {

constexpr int RBLOCK = 256;
constexpr int RRUN = 32;                   // consecutive kmers per lane (16 until late in round 2: the window cut is 45 instructions a run)
constexpr int RSYMS = RBLOCK * RRUN;       // source symbols staged per tile: one load round of RSYMS * SRC_BITS / 64 words, nothing more
// kmers per tile: 253 runs.  The last run's window (31 further starts, K - 1 <= 63 symbols of overlap) and up to 31 symbols of
// misalignment in front of the first fit into the RSYMS staged symbols, so a tile needs no extra staging pass for its overlap --
// a pass costs the wavefront the same 160 cycles whether 8 lanes or 64 have a word to recode (three lanes idle in the roll instead).
constexpr int RTILE = (RBLOCK - 3) * RRUN;
static_assert(RTILE + 63 + 31 <= RSYMS, "a tile's windows fit into one load round");
enum RunMode { RMODE_XOR = 0, RMODE_SKETCH = 1 };

// CANON: canonical kmers (always for the sketch); false = the forward kmers (kmers_reduce_xor with canonical = 0)
// HALF: one-word kmers of at most 32 bits (K <= 16), chosen by the host: their full runs work on the high halves of the windows
template <int SRC_BITS, int RMODE, int N = 1, bool CANON = true, bool HALF = false>
__global__ __launch_bounds__(RBLOCK) void run_kernel(const StreamArgs a) {
    static_assert(!HALF || N == 1, "HALF: one-word kmers");
    static_assert(RMODE != RMODE_SKETCH || CANON, "the sketch is over canonical kmers");
    __shared__ uint64_t lds[RSYMS * 2 / 64 + 8];  // the staged 2-bit stream + the window words past it (read, never used)
    __shared__ uint8_t lut[SRC_BITS == 8 ? 256 : 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t k = a.k;
    const uint64_t mask = head_mask((int)k, 2);
    if constexpr (SRC_BITS == 8) {
        for (uint32_t i = tid; i < 256u; i += RBLOCK) lut[i] = ascii_entry(a.ascii_table, i);
    }
    uint64_t threshold = a.threshold;
    // RMODE_SKETCH: a copy of the current bottom-s set, so that hashes it already holds are dropped here
    // instead of filling the candidate buffer -- in low-complexity sequence (poly-A, short tandem repeats)
    // the same few small hashes recur millions of times.
    extern __shared__ uint64_t sbest[];
    uint32_t nb = 0;
    if constexpr (RMODE == RMODE_SKETCH) {
        if (a.threshold_ptr) threshold = *a.threshold_ptr;  // uniform; constant for the whole launch
        if (a.best) {
            nb = (uint32_t)*a.best_n_ptr;
            for (uint32_t i = tid; i < nb; i += RBLOCK) sbest[i] = a.best[i];  // visible after the tile loop's first barrier
        }
    }
    uint64_t xacc = 0, xacc_left = 0;  // (xacc_left: XOR of LEFT-aligned one-word kmers, see the full-run path below)
    uint32_t xacc_left_hi = 0;         // (... of their high halves alone when K <= 16)

    struct Geo { uint64_t w0; uint32_t b0, nw, mt; };
    auto geometry = [&](uint64_t tile) {
        Geo g;
        const uint64_t m0 = tile * RTILE;
        const uint64_t left = a.n_kmers - m0;
        g.mt = left < (uint64_t)RTILE ? (uint32_t)left : (uint32_t)RTILE;
        const uint64_t bit0 = a.first_bit + m0 * SRC_BITS;
        g.w0 = bit0 >> 6;
        g.b0 = (uint32_t)(bit0 & 63u) / SRC_BITS;
        const uint64_t end_bit = bit0 + ((uint64_t)(g.mt - 1) + k) * SRC_BITS;
        g.nw = (uint32_t)(((end_bit + 63) >> 6) - g.w0);
        return g;
    };
    constexpr int PRE = RSYMS * SRC_BITS / 64 / RBLOCK;  // source words per thread per tile: 1, 2 or 4
    uint64_t pre[PRE];
    auto prefetch = [&](const Geo &g) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t wi = tid + (uint32_t)i * RBLOCK;
            pre[i] = wi < g.nw ? a.src[g.w0 + wi] : 0;
        }
    };

    // Candidates of a tile are collected in LDS and appended to the global buffer with ONE atomic on its
    // counter per tile: per-candidate atomics on that single word top out near 90 per microsecond, which
    // made the early rounds (tens of thousands of candidates in a short chunk) 5-10x slower than their
    // arithmetic (profiles/r01_tuning.md).
    constexpr uint32_t CB = RMODE == RMODE_SKETCH ? 512u : 1u;
    __shared__ uint64_t cbuf[CB];
    // the workgroup's own recently seen candidates (direct-mapped, plain loads/stores: a stale or lost
    // entry only costs a look at the global table): repeats of a homopolymer or tandem-repeat kmer stop here
    constexpr uint32_t LC = RMODE == RMODE_SKETCH ? 256u : 1u;
    __shared__ uint64_t lcache[LC];
    if constexpr (RMODE == RMODE_SKETCH) {
        for (uint32_t i = tid; i < LC; i += RBLOCK) lcache[i] = ~0ull;  // ~0 is never cached (see below)
    }
    __shared__ uint32_t ccount;
    __shared__ unsigned long long cbase;
    if (RMODE == RMODE_SKETCH && tid == 0) ccount = 0;  // visible after the tile loop's first barrier
    uint64_t prev_h = ~0ull;  // last candidate of this lane: a homopolymer run repeats one kmer
    auto candidate = [&](uint64_t h) {
        if (h < threshold && h != prev_h) {
            prev_h = h;
            uint32_t lo = 0, hi = nb;  // lower_bound(sbest, h)
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sbest[mid] < h) lo = mid + 1;
                else hi = mid;
            }
            bool fresh = lo == nb || sbest[lo] != h;
            if (fresh && h != ~0ull) {
                const uint32_t slot = (uint32_t)((h * 0x9E3779B97F4A7C15ull) >> 56) & (LC - 1u);
                if (lcache[slot] == h) fresh = false;
                else lcache[slot] = h;
            }
            if (fresh && sketch_is_new(a, h)) {
                const uint32_t p = atomicAdd(&ccount, 1u);
                if (p < CB) cbuf[p] = h;
                else sketch_append(a, h);  // the tile's LDS buffer is full: straight to the global one
            }
        }
    };
    // MinHash: a hash can only be a candidate if its HIGH half is at most the threshold's (for the bottom 1000 of a Gbase the
    // threshold is about 2^44: one kmer in a million passes).  The high half of the last fx_hash step, ((rotl(h, 5) ^ w) * C) >> 32,
    // is mul_hi(y_lo, C_lo) + y_lo * C_hi + y_hi * C_lo -- for kmers of at most 32 bits and seed 0 two multiplies and an add where
    // the full product, the 64-bit compare and the test against the previous candidate took twice that (round 5; the full hash is
    // computed for the survivors only, candidate() keeps the exact test).
    const uint32_t thr_hi = (uint32_t)(threshold >> 32);
    auto sketch_one = [&](const uint64_t (&c)[N]) {
        uint64_t h = a.seed;
#pragma unroll
        for (int w = 0; w + 1 < N; ++w) h = fx_step(h, c[w]);
        const uint64_t y = ((h << 5) | (h >> 59)) ^ c[N - 1];
        const uint32_t ylo = (uint32_t)y, yhi = (uint32_t)(y >> 32);
        const uint32_t hhi = __umulhi(ylo, (uint32_t)FX_CONSTANT) + ylo * (uint32_t)(FX_CONSTANT >> 32) + yhi * (uint32_t)FX_CONSTANT;
        if (hhi <= thr_hi) candidate(y * FX_CONSTANT);
    };
    // one kmer (N words, head first) and its reverse complement -> the consumer
    auto consume = [&](const uint64_t (&fw)[N], const uint64_t (&rc)[N]) {
        const bool lt = !CANON || kmer_less<N>(fw, rc);  // fw < rv ? fw : rv, CanonicalKmers.jl:220-225
        uint64_t c[N];
#pragma unroll
        for (int w = 0; w < N; ++w) c[w] = lt ? fw[w] : rc[w];
        if constexpr (RMODE == RMODE_XOR) xacc ^= c[0];            // the reducer of test/benchmark.jl:9-15: kmer.data[1]
        else sketch_one(c);                                         // kmer.jl:255-261
    };

#ifdef KMERS_STAMPS  // diagnostic builds: where a wavefront of this kernel spends its life (tools/run_stamps.py)
    const uint64_t st_rt0 = __builtin_amdgcn_s_memrealtime();
    uint64_t st_c[4] = {0, 0, 0, 0}, st_t = __builtin_amdgcn_s_memtime(), st_tiles = 0;
#define RSTAMP(i) do { const uint64_t n_ = __builtin_amdgcn_s_memtime(); st_c[i] += n_ - st_t; st_t = n_; } while (0)
#else
#define RSTAMP(i)
#endif
    uint64_t tile = blockIdx.x;
    if (tile < a.n_tiles) prefetch(geometry(tile));
    for (; tile < a.n_tiles; tile += gridDim.x) {
        const Geo g = geometry(tile);
        RSTAMP(3);     // (roll of the previous tile)
        block_sync();  // previous tile's readers are done with the stream
        RSTAMP(0);     // waiting for the workgroup
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const uint32_t wi = tid + (uint32_t)i * RBLOCK;
            if (wi < g.nw) {
                const uint64_t f = stage_word<SRC_BITS, 2>(lds, wi, pre[i], lut);
                if constexpr (SRC_BITS != 2) {
                    if (f) report_bad_symbols<SRC_BITS, true>(a.err_slot, a.first_bit, a.inspect_end, 1u, k, g.w0 + wi, f, pre[i], a.err_origin);
                }
            }
        }
        if (tile + gridDim.x < a.n_tiles) prefetch(geometry(tile + gridDim.x));
        RSTAMP(1);     // stage (incl. the wait for this tile's words)
        block_sync();
        RSTAMP(2);     // waiting for the workgroup's stage
#ifdef KMERS_STAMPS
        ++st_tiles;
#endif

        const uint32_t r0 = tid * RRUN;
        if (r0 < g.mt) {
            const uint32_t cnt = g.mt - r0 < (uint32_t)RRUN ? g.mt - r0 : (uint32_t)RRUN;
            const uint32_t bit = 2u * (r0 + g.b0);
            const uint32_t q = bit >> 6, s = bit & 63u;
            // (N + 1) * 64 stream bits from the run's first symbol; bits past the staged stream only reach kmers j >= cnt
            uint64_t W[N + 1];
            {
                uint64_t lo = lds[q];
#pragma unroll
                for (int j = 0; j <= N; ++j) {
                    const uint64_t hi = lds[q + j + 1];
                    W[j] = funnel64(lo, hi, s);
                    lo = hi;
                }
            }
            bool consumed = false;
            if constexpr (N == 1) {
                if (cnt == (uint32_t)RRUN) {
                    // A full run of one-word kmers, priced by profiles/r04_valu_rates.txt (everything but the simplest integer
                    // instructions costs a SIMD 4.1 cycles): the 32 kmers are WINDOWS of two 128-bit streams, cut with two
                    // v_alignbit each at shifts known at compile time -- no rolling recurrence, no masks.  Kmers are kept
                    // LEFT-aligned (first symbol in bits 62..63): the forward kmer j is bits [64 - 2j, 128 - 2j) of the
                    // symbol-reversed stream rev2(W0):rev2(W1) whatever K is, the reverse complement bits [2j, 2j + 64) of the
                    // complemented stream moved down by 2K bits once per run.  Below a kmer's 2K bits sit later symbols of the
                    // stream: they cannot change fw < rc unless the kmers are equal (then either is the canonical one) and they
                    // are shifted out where the kmer is used (XOR: once per lane at the end, shifts commute with XOR).
                    // 36 cycles of a SIMD per kmer where the rolling step takes 50-54 (tools/device_probes/roll_rate.hip).
                    const uint32_t up = 64u - 2u * k;  // 0..62
                    const uint64_t n0 = ~W[0], n1 = ~W[1];
                    const uint64_t A = n0 << up, B = ((n0 >> (2u * k - 1u)) >> 1) | (n1 << up);  // (~W0 : ~W1) >> 2K, zeros below
                    const uint64_t R0 = rev2(W[0]), R1 = rev2(W[1]);
                    const uint32_t rd[4] = {(uint32_t)R1, (uint32_t)(R1 >> 32), (uint32_t)R0, (uint32_t)(R0 >> 32)};
                    const uint32_t td[4] = {(uint32_t)A, (uint32_t)(A >> 32), (uint32_t)B, (uint32_t)(B >> 32)};
                    auto cut = [](const uint32_t (&d)[4], uint32_t bit) {  // 64 bits of the 128-bit stream from `bit` (<= 64) on
                        const uint32_t i = bit >> 5, sh = bit & 31u;
                        const uint32_t lo = sh ? __builtin_amdgcn_alignbit(d[i + 1], d[i], sh) : d[i];
                        const uint32_t hi = i + 2 < 4 ? (sh ? __builtin_amdgcn_alignbit(d[i + 2], d[i + 1], sh) : d[i + 1]) : d[i + 1] >> sh;
                        return ((uint64_t)hi << 32) | lo;
                    };
                    if constexpr (HALF) {
                        // kmers of at most 32 bits sit whole in the HIGH halves of the left-aligned windows: one
                        // v_alignbit per kmer and strand, and the canonical one is v_min_u32 -- 14 cycles of a SIMD per kmer for
                        // the XOR instead of 36.  (MinHash.jl's example sketches CanonicalDNAMers{16}, docs/src/minhash.md:34.)
                        auto cut_hi = [](const uint32_t (&d)[4], uint32_t bit) {  // bits [bit + 32, bit + 64) of the stream
                            const uint32_t i = bit >> 5, sh = bit & 31u;
                            return i + 2 < 4 ? (sh ? __builtin_amdgcn_alignbit(d[i + 2], d[i + 1], sh) : d[i + 1]) : d[i + 1] >> sh;
                        };
                        const uint32_t up32 = 32u - 2u * k;  // 0..30
#pragma unroll
                        for (uint32_t j = 0; j < (uint32_t)RRUN; ++j) {
                            const uint32_t f = cut_hi(rd, 64u - 2u * j), r = cut_hi(td, 2u * j);
                            const uint32_t c = CANON ? (f < r ? f : r) : f;
                            if constexpr (RMODE == RMODE_XOR) {
                                xacc_left_hi ^= c;
                            } else {
                                const uint64_t cr[1] = {(uint64_t)(c >> up32)};
                                sketch_one(cr);
                            }
                        }
                    } else {
#pragma unroll
                        for (uint32_t j = 0; j < (uint32_t)RRUN; ++j) {
                            const uint64_t f = cut(rd, 64u - 2u * j), r = cut(td, 2u * j);
                            const uint64_t c = (!CANON || f < r) ? f : r;  // fw < rv ? fw : rv, CanonicalKmers.jl:220-225
                            if constexpr (RMODE == RMODE_XOR) {
                                xacc_left ^= c;
                            } else {
                                const uint64_t cr[1] = {c >> up};
                                sketch_one(cr);
                            }
                        }
                    }
                    consumed = true;
                }
            }
            if (!consumed) {  // two-word kmers, and the last (short) run of a sequence: the rolling recurrence
            // first kmer of the run (see stream_kernel.hpp `window`): fw = symbol-reversed window, rc = complement
            uint64_t fw[N], rc[N];
            {
                uint64_t Wm[N], R[N];
#pragma unroll
                for (int j = 0; j < N; ++j) Wm[j] = W[j];
                Wm[N - 1] &= mask;
#pragma unroll
                for (int j = 0; j < N; ++j) {
                    rc[N - 1 - j] = ~Wm[j];
                    R[j] = rev2(Wm[j]);
                }
                rc[0] &= mask;
                const uint32_t sh = 64u * N - 2u * k;  // 0..62
                fw[0] = R[0] >> sh;
#pragma unroll
                for (int i = 1; i < N; ++i) fw[i] = (R[i] >> sh) | ((R[i - 1] << 1) << (63u - sh));
            }
            // the 31 symbols that enter afterwards: stream symbols K, K+1, ... of the run
            const uint32_t kb = 2u * k - 64u * (N - 1);  // bit offset of symbol K inside W[N-1] : W[N] (1..64)
            const uint64_t S = kb == 64u ? W[N] : funnel64(W[N - 1], W[N], kb);
            const uint32_t top = 2u * (k - 1u) - 64u * (N - 1);  // bit of the first symbol inside the head word
            consume(fw, rc);
            auto roll = [&](uint32_t j) {
                const uint64_t sym = (uint32_t)(S >> (2u * (j - 1u))) & 3u;
                // shift_encoding (construction_utils.jl:129-134) / shift_first_encoding of the complement (kmer.jl:511-518)
#pragma unroll
                for (int w = 0; w < N - 1; ++w) fw[w] = (fw[w] << 2) | (fw[w + 1] >> 62);
                fw[N - 1] = (fw[N - 1] << 2) | sym;
                fw[0] &= mask;
#pragma unroll
                for (int w = N - 1; w > 0; --w) rc[w] = (rc[w] >> 2) | (rc[w - 1] << 62);
                rc[0] = (rc[0] >> 2) | ((sym ^ 3u) << top);
                consume(fw, rc);
            };
            if (cnt == (uint32_t)RRUN) {
#pragma unroll
                for (uint32_t j = 1; j < (uint32_t)RRUN; ++j) roll(j);
            } else {
                for (uint32_t j = 1; j < cnt; ++j) roll(j);
            }
            }
        }
        if constexpr (RMODE == RMODE_SKETCH) {
            block_sync();
            const uint32_t filled = ccount < CB ? ccount : CB;  // uniform
            if (filled) {
                if (tid == 0) cbase = atomicAdd(reinterpret_cast<unsigned long long *>(a.out_b), (unsigned long long)filled);
                block_sync();
                for (uint32_t i = tid; i < filled; i += RBLOCK) {
                    const unsigned long long pos = cbase + i;
                    if (pos < a.capacity) a.out_a[pos] = cbuf[i];
                }
                block_sync();
                if (tid == 0) ccount = 0;  // ordered before the next tile's candidates by its two barriers
            }
        }
    }

#ifdef KMERS_STAMPS
    RSTAMP(3);
    if (a.stamps && (tid & 63u) == 0) {
        uint64_t *o = a.stamps + ((uint64_t)blockIdx.x * 4 + (tid >> 6)) * 8;
        o[0] = st_rt0;
        o[1] = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 4; ++i) o[2 + i] = st_c[i];
        o[6] = st_tiles;
    }
#endif
    if constexpr (RMODE == RMODE_XOR) {
        if constexpr (N == 1) xacc ^= (xacc_left ^ ((uint64_t)xacc_left_hi << 32)) >> (64u - 2u * k);
        // wavefront XOR-reduce (64 lanes), then one atomic per wave
        for (int off = 32; off > 0; off >>= 1) xacc ^= __shfl_xor(xacc, off, 64);
        if ((tid & 63u) == 0) atomicXor(reinterpret_cast<unsigned long long *>(a.out_a), (unsigned long long)xacc);
    }
}

}  // namespace kmers
