/* canonical_hashes.c -- the C ABI used from plain C (no Python, no Julia):
 *   CanonicalDNAMers{31}(seq) collected together with fx_hash of every element, for an ASCII
 *   sequence given on the command line (or a built-in one), then the bottom-8 MinHash sketch.
 *
 *   gcc -std=c99 -Iinclude examples/canonical_hashes.c -Lkmers.jl_amd/csrc -lkmers_hip \
 *       -Wl,-rpath,$PWD/kmers.jl_amd/csrc -o canonical_hashes && ./canonical_hashes ACGT...
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kmers_hip.h"

int main(int argc, char **argv) {
    const char *text = argc > 1 ? argv[1]
                                : "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCACGATCTTAGCACTTGCTAGGGATTCGAGGATC";
    const int k = 31;
    size_t len = strlen(text);
    /* ASCII sources are read in 8-byte words: give the library a padded, 8-byte aligned copy */
    size_t padded = (len + 15) / 8 * 8;
    uint64_t *bytes = calloc(padded / 8 + 1, 8);
    memcpy(bytes, text, len);

    kmers_ctx *ctx = NULL;
    if (kmers_ctx_create(0, NULL, &ctx) != KMERS_OK) {
        fprintf(stderr, "no usable HIP device (this library has no CPU fallback)\n");
        return 2;
    }
    kmers_seq seq = {bytes, len, 0, 0, 8 /* ASCII */, 0 /* DNA kmers */};
    uint64_t n = kmers_count(len, k, 1);
    uint64_t *kmers = malloc((n ? n : 1) * 8), *hashes = malloc((n ? n : 1) * 8);
    kmers_result res;
    int rc = kmers_canonical(ctx, &seq, k, 2, kmers, hashes, 0, KMERS_MEM_HOST, &res);
    if (rc == KMERS_E_ENCODE) {
        printf("EncodeError: cannot encode 0x%02x (Char '%c') at position %llu\n", res.err_enc, (char)res.err_enc,
               (unsigned long long)res.err_pos);
        return 1;
    }
    if (rc != KMERS_OK) {
        fprintf(stderr, "kmers_canonical: %d: %s\n", rc, kmers_last_error(ctx));
        return 2;
    }
    printf("%llu canonical %d-mers\n", (unsigned long long)res.n_out, k);
    for (uint64_t i = 0; i < res.n_out && i < 3; ++i)
        printf("  kmer[%llu] = 0x%016llx  fx_hash = 0x%016llx\n", (unsigned long long)i, (unsigned long long)kmers[i],
               (unsigned long long)hashes[i]);
    uint64_t sketch[8];
    rc = kmers_minhash(ctx, &seq, k, 2, 0, 8, sketch, KMERS_MEM_HOST, &res);
    if (rc == KMERS_OK) {
        printf("bottom-%llu MinHash sketch:", (unsigned long long)res.n_out);
        for (uint64_t i = 0; i < res.n_out; ++i) printf(" %016llx", (unsigned long long)sketch[i]);
        printf("\n");
    }
    kmers_ctx_destroy(ctx);
    free(kmers); free(hashes); free(bytes);
    return rc == KMERS_OK ? 0 : 2;
}
