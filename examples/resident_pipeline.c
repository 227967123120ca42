/* resident_pipeline.c -- a plain-C host that keeps its sequence and its outputs in HBM (no Python, no Julia, no HIP headers):
 *   CanonicalDNAMers{31}(seq) + fx_hash of every element (src/iterators/CanonicalKmers.jl:199-225, src/kmer.jl:255-261) over a
 *   synthetic LongDNA{4} sequence, launched asynchronously a few times,
 *     (a) into two plain device allocations (KMERS_PARAM_POOL = 0: kmers_dev_alloc = hipMalloc),
 *     (b) into two blocks of the device's CLASS POOL (the default: the arrays of a launch, allocated one after the other, lie in
 *         different region classes of HBM and are written at the two-class rate, include/kmers_hip.h),
 *     (c) like (b), but with FRESH outputs for every launch -- {alloc, alloc, launch, free, free}, what `collect` per sequence
 *         amounts to: a freed block stays mapped in the pool's cache and the next request of its shape takes it back,
 *   and prints the time per launch of each.  The runs must produce identical elements.
 *
 *   gcc -std=c99 -Iinclude examples/resident_pipeline.c -Lkmers.jl_amd/csrc -lkmers_hip \
 *       -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o resident_pipeline && ./resident_pipeline [Mbases]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "kmers_hip.h"

#define CHECK(call)                                                                       \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != KMERS_OK) {                                                            \
            fprintf(stderr, "%s: status %d: %s\n", #call, rc_, kmers_last_error(ctx));   \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)

static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec / 1e6;
}

int main(int argc, char **argv) {
    const uint64_t n_bases = (argc > 1 ? strtoull(argv[1], NULL, 10) : 256) * 1000000ull;
    const int k = 31, reps = 10;
    const uint64_t n = kmers_count(n_bases, k, 1), n_words = (n_bases * 4 + 63) / 64;
    enum { HEAD = 4096 };
    uint64_t head[2][2][HEAD];
    kmers_ctx *ctx = NULL;
    if (kmers_ctx_create(0, NULL, &ctx) != KMERS_OK) {
        fprintf(stderr, "no usable HIP device (this library has no CPU fallback)\n");
        return 2;
    }
    double ms[2] = {0, 0};
    for (int with_pool = 0; with_pool < 2; ++with_pool) {
        CHECK(kmers_ctx_set_param(ctx, KMERS_PARAM_POOL, with_pool));
        void *words = NULL, *out_kmers = NULL, *out_hashes = NULL;
        CHECK(kmers_dev_alloc(ctx, (n_words + 2) * 8, &words));
        CHECK(kmers_dev_alloc(ctx, n * 8, &out_kmers));   /* the arrays of one launch, one after the other */
        CHECK(kmers_dev_alloc(ctx, n * 8, &out_hashes));
        if (with_pool) {
            size_t n_chunks = 0, chunk_bytes = 0;
            unsigned char classes[2][64];
            void *arrays[2] = {out_kmers, out_hashes};
            for (int a = 0; a < 2; ++a) {
                CHECK(kmers_pool_layout(ctx, arrays[a], &chunk_bytes, classes[a], sizeof classes[a], &n_chunks));
                printf("pool: %s in %zu handle%s of %zu GiB, classes ", a ? "hashes" : "kmers", n_chunks, n_chunks == 1 ? "" : "s", chunk_bytes >> 30);
                for (size_t i = 0; i < n_chunks && i < sizeof classes[a]; ++i) putchar('A' + classes[a][i]);
                putchar('\n');
            }
        }
        CHECK(kmers_synth_dna(ctx, 42, 0, n_words, 4, 0, (uint64_t *)words));
        kmers_seq seq = {(const uint64_t *)words, n_bases, 0, 0, 4, 0};
        kmers_result res;
        for (int warm = 0; warm < 3; ++warm)
            CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)out_kmers, (uint64_t *)out_hashes, 0, KMERS_MEM_DEVICE | KMERS_ASYNC, &res));
        CHECK(kmers_sync(ctx, &res));
        const double t0 = now_ms();
        for (int r = 0; r < reps; ++r)
            CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)out_kmers, (uint64_t *)out_hashes, 0, KMERS_MEM_DEVICE | KMERS_ASYNC, &res));
        CHECK(kmers_sync(ctx, &res));
        ms[with_pool] = (now_ms() - t0) / reps;
        const uint64_t m = n < HEAD ? n : HEAD;
        CHECK(kmers_memcpy_d2h(ctx, head[with_pool][0], out_kmers, m * 8));
        CHECK(kmers_memcpy_d2h(ctx, head[with_pool][1], out_hashes, m * 8));
        for (uint64_t i = 0; i < m; ++i)
            if (head[with_pool][1][i] != head[with_pool][0][i] * 0x517cc1b727220a95ull) {
                fprintf(stderr, "hash %llu is not fx_hash of its kmer\n", (unsigned long long)i);
                return 1;
            }
        printf("%-24s %8.3f ms per launch = %6.1f Gbases/s = %5.2f TB/s of algorithmic traffic (16.5 B per kmer)\n",
               with_pool ? "outputs from the pool:" : "plain allocations:", ms[with_pool], n_bases / ms[with_pool] / 1e6,
               16.5 * n / ms[with_pool] / 1e9);
        if (with_pool) { /* (c) fresh outputs per launch: the collect loop */
            CHECK(kmers_dev_free(ctx, out_kmers));
            CHECK(kmers_dev_free(ctx, out_hashes));
            const double t1 = now_ms();
            for (int r = 0; r < reps; ++r) {
                CHECK(kmers_dev_alloc(ctx, n * 8, &out_kmers));
                CHECK(kmers_dev_alloc(ctx, n * 8, &out_hashes));
                CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)out_kmers, (uint64_t *)out_hashes, 0, KMERS_MEM_DEVICE | KMERS_ASYNC, &res));
                if (r + 1 < reps) {
                    CHECK(kmers_dev_free(ctx, out_kmers));   /* no wait: the next user of the block comes after this launch */
                    CHECK(kmers_dev_free(ctx, out_hashes));
                }
            }
            CHECK(kmers_sync(ctx, &res));
            const double fresh = (now_ms() - t1) / reps;
            uint64_t stats[KMERS_POOL_STATS];
            CHECK(kmers_pool_stats(ctx, stats, KMERS_POOL_STATS));
            printf("%-24s %8.3f ms per {alloc, alloc, launch, free, free} (%llu of the pool's %llu allocations came from its cache)\n",
                   "fresh outputs per launch:", fresh, (unsigned long long)stats[4], (unsigned long long)(stats[4] + stats[5]));
            CHECK(kmers_memcpy_d2h(ctx, head[0][1], out_hashes, m * 8));
            if (memcmp(head[0][1], head[1][1], m * 8) != 0) {
                fprintf(stderr, "the launch into fresh outputs differs\n");
                return 1;
            }
        }
        CHECK(kmers_dev_free(ctx, words));
        CHECK(kmers_dev_free(ctx, out_kmers));
        CHECK(kmers_dev_free(ctx, out_hashes));
    }
    const uint64_t m = n < HEAD ? n : HEAD;
    if (memcmp(head[0], head[1], sizeof head[0]) != 0 && m == HEAD) {
        fprintf(stderr, "the two runs differ\n");
        return 1;
    }
    printf("both runs: %llu elements, the first %llu identical: equal\n", (unsigned long long)n, (unsigned long long)m);
    /* A launch with ONE output array (the kmers without their hashes): an ordinary block of the pool against a block taken by
     * role, whose second half lies in another region class than its first and which is written through two windows
     * (include/kmers_hip.h; arrays of 2 GiB and more: a block is made of whole 1 GiB handles). */
    for (int role = KMERS_ALLOC_DEFAULT; role <= KMERS_ALLOC_LONE_OUTPUT; ++role) {
        void *words = NULL, *only_out = NULL;
        CHECK(kmers_dev_alloc_role(ctx, n * 8, role, &only_out));
        CHECK(kmers_dev_alloc(ctx, (n_words + 2) * 8, &words));
        CHECK(kmers_synth_dna(ctx, 42, 0, n_words, 4, 0, (uint64_t *)words));
        kmers_seq seq = {(const uint64_t *)words, n_bases, 0, 0, 4, 0};
        kmers_result res;
        for (int warm = 0; warm < 3; ++warm)
            CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)only_out, NULL, 0, KMERS_MEM_DEVICE | KMERS_ASYNC, &res));
        CHECK(kmers_sync(ctx, &res));
        const double t0 = now_ms();
        for (int r = 0; r < reps; ++r)
            CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)only_out, NULL, 0, KMERS_MEM_DEVICE | KMERS_ASYNC, &res));
        CHECK(kmers_sync(ctx, &res));
        const double t = (now_ms() - t0) / reps;
        CHECK(kmers_memcpy_d2h(ctx, head[1][1], only_out, m * 8));
        if (memcmp(head[1][1], head[1][0], m * 8) != 0) {
            fprintf(stderr, "the kmers-only launch differs from the kmers of the two-output launch\n");
            return 1;
        }
        printf("%-24s %8.3f ms per launch = %5.2f TB/s (8.5 B per kmer)\n", role ? "kmers only, by role:" : "kmers only, plain block:", t,
               8.5 * n / t / 1e9);
        CHECK(kmers_dev_free(ctx, words));
        CHECK(kmers_dev_free(ctx, only_out));
    }
    kmers_ctx_destroy(ctx);
    return 0;
}
