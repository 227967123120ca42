/* sharded_canonical.c -- the sharded path of the C ABI from plain C (no Python, no Julia, no torch): one long synthetic
 * LongDNA{2} cut into S contiguous shards (kmers_shard_plan), every shard run as the rank that owns it would run it:
 * own words in HBM, the (K-1)-base halo fetched through kmers_comm_sendrecv on an RCCL communicator, then
 * CanonicalDNAMers{31} with index_origin = the shard's first base.  One process owns one GPU here, so the communicator has
 * one rank and the "neighbour" of a shard is another buffer of the same rank (peer = self); with one process per GPU the
 * same calls are kmers_halo_exchange(ctx, comm, &shard, words).  The XOR of all canonical kmers over the shards must
 * equal that of the unsharded sequence (kmers_reduce_xor), and the shards' element counts go through
 * kmers_offsets_allgather / kmers_first_error_allreduce.
 *
 *   gcc -std=c99 -Iinclude examples/sharded_canonical.c -Lkmers.jl_amd/csrc -lkmers_hip \
 *       -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o sharded_canonical && ./sharded_canonical 8 10000000
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmers_hip.h"

#define CHECK(call)                                                                        \
    do {                                                                                   \
        int rc_ = (call);                                                                  \
        if (rc_ != KMERS_OK) {                                                             \
            fprintf(stderr, "%s: status %d: %s\n", #call, rc_, kmers_last_error(ctx));    \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

int main(int argc, char **argv) {
    const int n_shards = argc > 1 ? atoi(argv[1]) : 8;
    const uint64_t n_bases = argc > 2 ? strtoull(argv[2], NULL, 10) : 10000000ull;
    const int k = 31, bits = 2;
    kmers_ctx *ctx = NULL;
    if (kmers_ctx_create(0, NULL, &ctx) != KMERS_OK) {
        fprintf(stderr, "no usable HIP device (this library has no CPU fallback)\n");
        return 2;
    }
    unsigned char id[KMERS_COMM_ID_BYTES];
    void *comm = NULL;
    CHECK(kmers_comm_id(id));
    CHECK(kmers_comm_create(ctx, id, 1, 0, &comm));
    int rank = -1, n_ranks = -1;
    CHECK(kmers_comm_rank(ctx, comm, &rank, &n_ranks));
    printf("communicator: rank %d of %d\n", rank, n_ranks);

    /* the whole sequence once (the reference result), resident */
    const uint64_t total_words = (n_bases * bits + 63) / 64;
    void *whole = NULL;
    CHECK(kmers_dev_alloc(ctx, (total_words + 2) * 8, &whole));
    CHECK(kmers_synth_dna(ctx, 0x5eed, 0, total_words, bits, 0, (uint64_t *)whole));
    kmers_seq all = {(const uint64_t *)whole, n_bases, 0, 0, bits, 0};
    kmers_result res;
    uint64_t want = 0;
    CHECK(kmers_reduce_xor(ctx, &all, k, 2, 1, &want, KMERS_MEM_DEVICE, &res));

    uint64_t got = 0, kmers_seen = 0;
    for (int g = 0; g < n_shards; ++g) {
        kmers_shard sh, next;
        CHECK(kmers_shard_plan(n_bases, k, 1, bits, n_shards, g, &sh));
        if (sh.n_kmers == 0) continue;
        /* this shard's buffer: ONLY its own words (generated in place), room for the halo behind them */
        void *buf = NULL, *out = NULL;
        CHECK(kmers_dev_alloc(ctx, (sh.n_own_words + sh.halo_words + 2) * 8, &buf));
        CHECK(kmers_synth_dna(ctx, 0x5eed, sh.first_word, sh.n_own_words, bits, 0, (uint64_t *)buf));
        if (sh.halo_words) {
            /* the first halo_words words of shard g+1, as its owner would send them (send_words of that shard) */
            CHECK(kmers_shard_plan(n_bases, k, 1, bits, n_shards, g + 1, &next));
            if (next.send_words != sh.halo_words) {
                fprintf(stderr, "plan mismatch\n");
                return 2;
            }
            CHECK(kmers_comm_sendrecv(ctx, comm, (const uint64_t *)whole + next.first_word, next.send_words, rank,
                                      (uint64_t *)buf + sh.n_own_words, sh.halo_words, rank));
        }
        CHECK(kmers_halo_exchange(ctx, comm, &sh, (uint64_t *)buf)); /* one rank: nothing to exchange, must succeed */
        CHECK(kmers_dev_alloc(ctx, sh.n_kmers * 8, &out));
        kmers_seq seq = {(const uint64_t *)buf, sh.n_bases, 0, sh.first_base, bits, 0};
        CHECK(kmers_canonical(ctx, &seq, k, 2, (uint64_t *)out, NULL, 0, KMERS_MEM_DEVICE, &res));
        uint64_t part = 0;
        CHECK(kmers_reduce_xor(ctx, &seq, k, 2, 1, &part, KMERS_MEM_DEVICE, &res));
        /* the materialised kmers of the shard, folded on the host */
        uint64_t *host = malloc(sh.n_kmers * 8), fold = 0;
        CHECK(kmers_memcpy_d2h(ctx, host, out, sh.n_kmers * 8));
        for (uint64_t i = 0; i < sh.n_kmers; ++i) fold ^= host[i];
        free(host);
        if (fold != part) {
            fprintf(stderr, "shard %d: materialised kmers and fused reducer disagree\n", g);
            return 1;
        }
        got ^= part;
        kmers_seen += sh.n_kmers;
        uint64_t off = 0, tot = 0;
        CHECK(kmers_offsets_allgather(ctx, comm, sh.n_kmers, &off, &tot));
        if (off != 0 || tot != sh.n_kmers) return 1; /* one rank: its own count */
        kmers_result first = {KMERS_OK, 0, 0, 0};
        CHECK(kmers_first_error_allreduce(ctx, comm, &first));
        kmers_dev_free(ctx, out);
        kmers_dev_free(ctx, buf);
    }
    printf("%d shards, %llu kmers, xor %016llx, unsharded %016llx: %s\n", n_shards, (unsigned long long)kmers_seen,
           (unsigned long long)got, (unsigned long long)want, got == want && kmers_seen == kmers_count(n_bases, k, 1) ? "equal" : "DIFFERENT");
    kmers_dev_free(ctx, whole);
    kmers_comm_destroy(ctx, comm);
    kmers_ctx_destroy(ctx);
    return got == want ? 0 : 1;
}
