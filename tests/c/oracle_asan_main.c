/* AddressSanitizer driver for the C oracle (CPU only; tests/test_oracle_sanitized.py builds and runs it).
 * Every buffer is an exact-size heap allocation, so a read or write one word past what the interface
 * promises (ceil(len * bps / 64) source words, n * N output words) aborts the run. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/kmers_oracle.h"

static uint64_t *exact(size_t words) {
    uint64_t *p = (uint64_t *)malloc(words ? words * 8 : 1);
    if (!p) exit(2);
    return p;
}

int main(void) {
    static const int KS[] = {1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 128};
    static const uint64_t LENS[] = {0, 1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 4099};
    unsigned long checks = 0;
    for (int src = 2; src <= 8; src *= 2) {
        for (int dst = 2; dst <= 4; dst *= 2) {
            for (size_t ki = 0; ki < sizeof KS / sizeof *KS; ++ki) {
                const int K = KS[ki];
                const int N = orc_n_coding_elements(K, dst);
                if (N > 4) continue;
                for (size_t li = 0; li < sizeof LENS / sizeof *LENS; ++li) {
                    const uint64_t L = LENS[li];
                    const size_t nw = (size_t)((L * (uint64_t)src + 63) / 64);
                    uint64_t *seq = exact(nw);
                    if (src == 8) {
                        for (uint64_t i = 0; i < nw * 8; ++i) ((unsigned char *)seq)[i] = "ACGTacgt"[(i * 7 + K) & 7];
                    } else {
                        orc_synth_words(1234 + K, 0, nw, src, 0, seq);
                    }
                    const uint64_t n = L >= (uint64_t)K ? L - (uint64_t)K + 1 : 0;
                    const int osrc = src == 8 ? 8 : src;  /* 8 = ASCII bytes, DNA */
                    orc_result res;
                    uint64_t *a = exact((size_t)n * N), *b = exact((size_t)n * N), *h = exact((size_t)n);
                    orc_fw_kmers(seq, L, osrc, dst, K, a, &res);
                    orc_fwrv(seq, L, osrc, dst, K, a, b, &res);
                    orc_canonical(seq, L, osrc, dst, K, a, h, 5, &res);
                    (void)orc_reduce_xor_canonical(seq, L, osrc, dst, K, &res);
                    for (int J = 1; J <= 70; J += 23) {
                        const uint64_t m = L >= (uint64_t)K ? (L - (uint64_t)K) / (uint64_t)J + 1 : 0;
                        uint64_t *s = exact((size_t)m * N);
                        orc_spaced(seq, L, osrc, dst, K, J, s, &res);
                        free(s);
                        ++checks;
                    }
                    if (dst == 2 && K <= 64) {
                        int64_t *st = (int64_t *)exact((size_t)n);
                        orc_unambiguous(seq, L, osrc, K, a, st, &res);
                        free(st);
                    }
                    if (L >= (uint64_t)K + 4) {
                        const int W = 5, stride = 3;
                        const uint64_t m = (L - (uint64_t)(K + W - 1)) / stride + 1;
                        uint64_t *mn = exact((size_t)m * N);
                        orc_minimizers(seq, L, osrc, dst, K, W, stride, 0, mn, &res);
                        orc_minimizers(seq, L, osrc, dst, K, W, stride, 1, mn, &res);
                        free(mn);
                    }
                    /* element-wise functions on the first kmer, exact-size in and out */
                    if (n && src != 8) {
                        uint64_t *x = exact((size_t)N), *y = exact((size_t)N);
                        memcpy(x, a, (size_t)N * 8);
                        orc_reverse(x, K, dst, y);
                        orc_complement(x, K, dst, y);
                        orc_reverse_complement(x, K, dst, y);
                        orc_canonical_kmer(x, K, dst, y);
                        (void)orc_iscanonical(x, K, dst);
                        (void)orc_fx_hash(x, N, 9);
                        orc_longseq_from_kmer(x, K, dst, y);
                        free(x);
                        free(y);
                    }
                    free(a);
                    free(b);
                    free(h);
                    free(seq);
                    ++checks;
                }
            }
        }
    }
    printf("asan driver: %lu cases, no invalid access\n", checks);
    return 0;
}
