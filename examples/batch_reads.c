/* batch_reads.c -- many short records in one launch from plain C:
 *   CanonicalDNAMers{21}(read) + fx_hash for every read of a small FASTA-like batch (reads with N are not an
 *   error here: KMERS_BATCH_SKIP marks the windows over them), and one MinHash sketch per read.
 *
 *   gcc -std=c99 -Iinclude examples/batch_reads.c -Lkmers.jl_amd/csrc -lkmers_hip \
 *       -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o batch_reads && ./batch_reads
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmers_hip.h"

int main(void) {
    static const char *reads[] = {
        "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCACGATCTTAGCAC",
        "ACGTACGTNNACGTACGTACGTACGTTTGACCAGTAGGACCATTAGA", /* two ambiguous calls */
        "ACGT",                                            /* shorter than K: yields nothing */
        "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG",
    };
    enum { N_READS = sizeof reads / sizeof *reads, K = 21, S = 8 };
    /* the pool: all reads back to back as ASCII bytes (8-byte aligned, padded); one span per read */
    size_t total = 0;
    for (int i = 0; i < N_READS; ++i) total += strlen(reads[i]);
    uint64_t *pool = calloc(total / 8 + 2, 8);
    kmers_span spans[N_READS];
    size_t pos = 0;
    for (int i = 0; i < N_READS; ++i) {
        size_t len = strlen(reads[i]);
        memcpy((char *)pool + pos, reads[i], len);
        spans[i].first_base = pos;
        spans[i].n_bases = len;
        pos += len;
    }
    kmers_ctx *ctx = NULL;
    if (kmers_ctx_create(0, NULL, &ctx) != KMERS_OK) {
        fprintf(stderr, "no usable HIP device (this library has no CPU fallback)\n");
        return 2;
    }
    kmers_seq seq = {pool, total, 0, 0, 8 /* ASCII */, 0 /* DNA kmers */};
    kmers_result res;
    uint64_t offsets[N_READS + 1];
    const int flags = KMERS_MEM_HOST | KMERS_BATCH_SKIP;
    /* size query, then the elements */
    if (kmers_batch(ctx, &seq, spans, N_READS, KMERS_BATCH_CANONICAL, K, 2, NULL, NULL, 0, offsets, 0, flags, &res) != KMERS_OK) {
        fprintf(stderr, "kmers_batch: %s\n", kmers_last_error(ctx));
        return 1;
    }
    uint64_t n = res.n_out;
    uint64_t *kmers = malloc((n ? n : 1) * 8), *hashes = malloc((n ? n : 1) * 8);
    if (kmers_batch(ctx, &seq, spans, N_READS, KMERS_BATCH_CANONICAL, K, 2, kmers, hashes, 0, offsets, n, flags, &res) != KMERS_OK) {
        fprintf(stderr, "kmers_batch: %s\n", kmers_last_error(ctx));
        return 1;
    }
    uint64_t sketches[N_READS * S], counts[N_READS];
    if (kmers_minhash_batch(ctx, &seq, spans, N_READS, K, 2, 0, S, sketches, counts, flags, &res) != KMERS_OK) {
        fprintf(stderr, "kmers_minhash_batch: %s\n", kmers_last_error(ctx));
        return 1;
    }
    for (int i = 0; i < N_READS; ++i) {
        uint64_t lo = offsets[i], hi = offsets[i + 1], skipped = 0;
        for (uint64_t g = lo; g < hi; ++g) skipped += kmers[g] == ~0ull;
        printf("read %d: %llu canonical %d-mers (%llu over an ambiguous call), sketch of %llu:", i, (unsigned long long)(hi - lo), K,
               (unsigned long long)skipped, (unsigned long long)counts[i]);
        for (uint64_t j = 0; j < counts[i]; ++j) printf(" %016llx", (unsigned long long)sketches[i * S + j]);
        printf("\n");
    }
    free(kmers);
    free(hashes);
    free(pool);
    kmers_ctx_destroy(ctx);
    return 0;
}
