/*
 * kmers_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See kmers_oracle.h for the scope, the parity pin and the usage rule.
 *
 * Step-for-step restatement of BioJulia/Kmers.jl v1.2.0; each function names
 * the reference lines it follows.  Julia semantics that differ from C are
 * restated explicitly:
 *   - left_shift/right_shift mask the count to 6 bits (tuple_bitflipping.jl:3-9)
 *   - Julia's plain `<<` gives 0 for counts >= 64 (used by get_mask, kmer.jl:603-605)
 *   - trailing_zeros(0) == 64
 * Third-party arithmetic that is NOT in the reference tree (BioSequences.jl
 * ~3.4.1/3.5, BioSymbols 5.1.3) is restated from its published behaviour:
 * extract_encoded_element, reversebits, complement_bitpar, the 2-bit (A,C,G,T =
 * 0..3) and 4-bit (one-hot A=1,C=2,G=4,T=8) nucleotide encodings.
 */
#include "kmers_oracle.h"

#include <string.h>

#define INL static inline __attribute__((always_inline))

/* ------------------------------------------------------------------------ */
/* src/tuple_bitflipping.jl:3-19                                             */
INL uint64_t left_shift(uint64_t x, int64_t n) { return x << (n & 63); }
INL uint64_t right_shift(uint64_t x, int64_t n) { return x >> (n & 63); }
INL uint64_t left_carry(uint64_t x, int64_t n) { return right_shift(x, 64 - n); }
INL uint64_t right_carry(uint64_t x, int64_t n) { return left_shift(x, 64 - n); }

/* src/tuple_bitflipping.jl:24-33,49 -- the recursion visits the tail first, so
 * the carry argument enters at the LAST word and travels towards the head. */
INL uint64_t leftshift_carry(uint64_t *x, const int N, int64_t nbits, uint64_t carry) {
    for (int i = N - 1; i >= 0; --i) {
        uint64_t new_head = left_shift(x[i], nbits) | carry;
        carry = left_carry(x[i], nbits);
        x[i] = new_head;
    }
    return carry;
}

/* src/tuple_bitflipping.jl:35-46,50 -- head first; the carry handed down is the
 * UNSHIFTED low bits, moved to the top by right_carry in the next word. */
INL uint64_t rightshift_carry(uint64_t *x, const int N, int64_t nbits, uint64_t carry) {
    for (int i = 0; i < N; ++i) {
        uint64_t new_head = right_shift(x[i], nbits) | right_carry(carry, nbits);
        uint64_t mask = left_shift((uint64_t)1, nbits) - 1;
        carry = x[i] & mask;
        x[i] = new_head;
    }
    return carry;
}

uint64_t orc_leftshift_carry(uint64_t *x, int N, int64_t nbits, uint64_t carry) {
    return leftshift_carry(x, N, nbits, carry);
}
uint64_t orc_rightshift_carry(uint64_t *x, int N, int64_t nbits, uint64_t carry) {
    return rightshift_carry(x, N, nbits, carry);
}

/* ------------------------------------------------------------------------ */
/* geometry: src/kmer.jl:103, :117-137                                       */
INL int n_coding_elements(int K, int bps) { return (K * bps + 63) / 64; } /* :123-125 */
INL int per_word_capacity(int bps) { return 64 / bps; }                    /* :127-129 */
INL int n_unused(int K, int bps) { return per_word_capacity(bps) * n_coding_elements(K, bps) - K; } /* :119,:131-133 */
INL int bits_unused(int K, int bps) { return n_unused(K, bps) * bps; }     /* :120-121 */
INL int elements_in_head(int K, int bps) { return per_word_capacity(bps) - n_unused(K, bps); } /* :135-137 */
/* src/kmer.jl:603-605 -- `UInt(1) << (64 - bits_unused) - 1` with Julia's `<<` (>= 64 -> 0) */
INL uint64_t get_mask(int K, int bps) {
    int sh = 64 - bits_unused(K, bps);
    uint64_t one_shifted = sh >= 64 ? (uint64_t)0 : ((uint64_t)1 << sh);
    return one_shifted - 1;
}
INL int trailing_zeros64(uint64_t x) { return x ? __builtin_ctzll(x) : 64; }
INL int count_ones64(uint64_t x) { return __builtin_popcountll(x); }

int orc_n_coding_elements(int K, int bps) { return n_coding_elements(K, bps); }
int orc_bits_unused(int K, int bps) { return bits_unused(K, bps); }
int orc_elements_in_head(int K, int bps) { return elements_in_head(K, bps); }
uint64_t orc_get_mask(int K, int bps) { return get_mask(K, bps); }

/* ------------------------------------------------------------------------ */
/* BioSequences (absent dependency), restated                                */

/* extract_encoded_element(seq::LongSequence, i): call sites FwKmers.jl:91,99,111;
 * CanonicalKmers.jl:101,114,125,138; construction_utils.jl:35,49,65 */
INL uint64_t extract_encoded_element(const uint64_t *data, uint64_t i, const int bps) {
    uint64_t bit = (i - 1) * (uint64_t)bps;
    return (data[bit >> 6] >> (bit & 63)) & (((uint64_t)1 << bps) - 1);
}

/* reversebits(x, BitsPerSymbol{bps}): reverse the order of the bps-bit symbols
 * of a word (call sites transformations.jl:7, construction.jl:216,305,320,323) */
INL uint64_t reversebits(uint64_t x, const int bps) {
    x = __builtin_bswap64(x);
    if (bps <= 4) x = ((x & 0xF0F0F0F0F0F0F0F0ull) >> 4) | ((x & 0x0F0F0F0F0F0F0F0Full) << 4);
    if (bps <= 2) x = ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
    return x;
}

/* complement_bitpar (call sites transformations.jl:16,23):
 * 2-bit: A<->T, C<->G is bitwise NOT.  4-bit: reverse the bits inside each
 * nibble (A=0001<->T=1000, C=0010<->G=0100; gap and N are fixed points). */
INL uint64_t complement_bitpar(uint64_t x, const int bps) {
    if (bps == 2) return ~x;
    x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
    x = ((x & 0xCCCCCCCCCCCCCCCCull) >> 2) | ((x & 0x3333333333333333ull) << 2);
    return x;
}

/* complement(::DNA) of one 4-bit symbol (CanonicalKmers.jl:116-117) */
INL uint64_t complement_nibble(uint64_t e) {
    return ((e & 1) << 3) | ((e & 2) << 1) | ((e & 4) >> 1) | ((e & 8) >> 3);
}

/* ------------------------------------------------------------------------ */
/* src/construction_utils.jl:129-134                                         */
INL void shift_encoding(uint64_t *data, const int N, const int K, const int bps, uint64_t enc) {
    if (K == 0) return; /* isempty(kmer) && return kmer */
    leftshift_carry(data, N, bps, enc);
    data[0] &= get_mask(K, bps);
}

/* src/kmer.jl:511-518 */
INL void shift_first_encoding(uint64_t *data, const int N, const int K, const int bps, uint64_t enc) {
    if (K == 0) return;
    rightshift_carry(data, N, bps, 0);
    data[0] |= left_shift(enc, (int64_t)(elements_in_head(K, bps) - 1) * bps);
}

void orc_shift_encoding(uint64_t *kmer, int K, int bps, uint64_t enc) {
    shift_encoding(kmer, n_coding_elements(K, bps), K, bps, enc);
}
void orc_shift_first_encoding(uint64_t *kmer, int K, int bps, uint64_t enc) {
    shift_first_encoding(kmer, n_coding_elements(K, bps), K, bps, enc);
}

/* src/construction.jl:108-110 throw_uncertain -> EncodeError(A, symbol) */
INL int throw_uncertain(orc_result *res, uint64_t pos, uint64_t enc) {
    res->status = ORC_E_ENCODE;
    res->err_pos = pos;
    res->err_enc = (uint32_t)(enc & 0xff);
    return ORC_E_ENCODE;
}

/* ---- AsciiEncode sources (src_bps codes ORC_SRC_ASCII_DNA / _RNA) -----------------------------
 * BioSequences.ascii_encode(A, byte) restated (absent dependency): the byte's symbol in the KMER's
 * alphabet A, or 0x80 when the byte is not a symbol of A.  2-bit: ACGT (DNA) / ACGU (RNA), either
 * case.  4-bit: the IUPAC letters "-ACMGRSVTWYHKDBN" (T->U for RNA), either case, value = the
 * one-hot/OR encoding.  Call sites: FwKmers.jl:123, CanonicalKmers.jl:156, construction_utils.jl:81,229 */
static uint8_t ascii_encode(int dst_bps, int rna, uint8_t byte) {
    static const char iupac[] = "-ACMGRSVTWYHKDBN";
    uint8_t c = byte;
    if (c >= 'a' && c <= 'z') c = (uint8_t)(c - 32);
    if (rna) {
        if (c == 'T') return 0x80;
        if (c == 'U') c = 'T';
    } else if (c == 'U') {
        return 0x80;
    }
    if (byte >= 0x80) return 0x80;
    if (dst_bps == 2) {
        switch (c) {
            case 'A': return 0;
            case 'C': return 1;
            case 'G': return 2;
            case 'T': return 3;
            default: return 0x80;
        }
    }
    for (int v = 0; v < 16; ++v)
        if (c == (uint8_t)iupac[v]) return (uint8_t)v;
    return 0x80;
}

/* ASCII_SKIPPING_LUT of the reference itself (src/iterators/common.jl:22-32): 0..3 for
 * "Aa","cC","gG","TtUu"; 0xf0 for "-MRSVWYHKDBN" either case; 0xff otherwise. */
static uint8_t ascii_skipping_lut(uint8_t byte) {
    switch (byte) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': case 'U': case 'u': return 3;
        default: break;
    }
    static const char amb[] = "-MRSVWYHKDBN";
    for (int i = 0; amb[i]; ++i)
        if (byte == (uint8_t)amb[i] || (amb[i] != '-' && byte == (uint8_t)(amb[i] + 32))) return 0xf0;
    return 0xff;
}

void orc_ascii_tables(int dst_bps, int rna, uint8_t *encode_lut, uint8_t *skipping_lut) {
    for (int b = 0; b < 256; ++b) {
        if (encode_lut) encode_lut[b] = ascii_encode(dst_bps, rna, (uint8_t)b);
        if (skipping_lut) skipping_lut[b] = ascii_skipping_lut((uint8_t)b);
    }
}

#define IS_ASCII(src_bps) ((src_bps) == ORC_SRC_ASCII_DNA || (src_bps) == ORC_SRC_ASCII_RNA)
#define IS_BYTES(src_bps) (IS_ASCII(src_bps) || (src_bps) == ORC_SRC_SYMBOLS)

/* One checked/recoded symbol fetch, per RecodingScheme (src/construction.jl:75-100):
 * same width -> Copyable, 4->2 FourToTwo, 2->4 TwoToFour.
 * Returns 0 and the encoding to shift in, or ORC_E_ENCODE. */
INL int fetch_recoded(const uint64_t *seq, uint64_t i, const int src_bps, const int dst_bps,
                      uint64_t *enc_out, orc_result *res) {
    if (src_bps == ORC_SRC_SYMBOLS) {
        /* GenericRecoding: symbol = convert(eltype(kmer), seq[i]) keeps the 4-bit value (DNA <-> RNA share encodings);
         * BioSequences.encode(A, symbol): a 2-bit alphabet takes only the four one-hot values (trailing_zeros), anything
         * else is EncodeError(A, symbol); a 4-bit alphabet takes every nucleotide value, gap included
         * (construction_utils.jl:98-100, kmer.jl:445-448) */
        uint8_t byte = ((const uint8_t *)seq)[i - 1];
        if (byte > 0x0f) return throw_uncertain(res, i, byte); /* not a nucleotide value at all */
        if (dst_bps == 4) {
            *enc_out = byte;
            return 0;
        }
        if (count_ones64(byte) != 1) return throw_uncertain(res, i, byte);
        *enc_out = (uint64_t)trailing_zeros64(byte);
        return 0;
    }
    if (IS_ASCII(src_bps)) { /* AsciiEncode: FwKmers.jl:117-129, construction_utils.jl:71-88, :220-236 */
        uint8_t byte = ((const uint8_t *)seq)[i - 1];
        uint8_t encoding = ascii_encode(dst_bps, src_bps == ORC_SRC_ASCII_RNA, byte);
        if (encoding > 0x7f) return throw_uncertain(res, i, byte); /* EncodeError(A, byte) */
        *enc_out = encoding;
        return 0;
    }
    uint64_t encoding = extract_encoded_element(seq, i, src_bps);
    if (src_bps == dst_bps) { /* Copyable: construction_utils.jl:65, FwKmers.jl:91 */
        *enc_out = encoding;
    } else if (src_bps == 2) { /* TwoToFour: construction_utils.jl:35, FwKmers.jl:99 */
        *enc_out = left_shift((uint64_t)1, (int64_t)encoding);
    } else { /* FourToTwo: construction_utils.jl:49-51, FwKmers.jl:111-113 */
        if (count_ones64(encoding) != 1) return throw_uncertain(res, i, encoding);
        *enc_out = (uint64_t)trailing_zeros64(encoding);
    }
    return 0;
}

/* src/construction_utils.jl:27-69 (TwoToFour, FourToTwo, Copyable) */
INL int unsafe_extract(const uint64_t *seq, const int src_bps, const int dst_bps, const int N,
                       const int K, uint64_t from, uint64_t *data, orc_result *res) {
    for (int w = 0; w < N; ++w) data[w] = 0; /* zero_tuple(T) */
    for (uint64_t i = from; i < from + (uint64_t)K; ++i) {
        uint64_t enc;
        if (fetch_recoded(seq, i, src_bps, dst_bps, &enc, res)) return ORC_E_ENCODE;
        leftshift_carry(data, N, dst_bps, enc);
    }
    return 0;
}

int orc_unsafe_extract(const uint64_t *seq, int src_bps, int dst_bps, int K, uint64_t from,
                       uint64_t *out, orc_result *res) {
    memset(res, 0, sizeof *res);
    return unsafe_extract(seq, src_bps, dst_bps, n_coding_elements(K, dst_bps), K, from, out, res);
}

/* src/construction_utils.jl:175-218 */
INL int unsafe_shift_from(const uint64_t *seq, const int src_bps, const int dst_bps, const int N,
                          const int K, uint64_t from, int S, uint64_t *data, orc_result *res) {
    for (int i = 0; i < S; ++i) {
        uint64_t enc;
        if (fetch_recoded(seq, from + (uint64_t)i, src_bps, dst_bps, &enc, res)) return ORC_E_ENCODE;
        shift_encoding(data, N, K, dst_bps, enc);
    }
    return 0;
}

int orc_unsafe_shift_from(const uint64_t *seq, int src_bps, int dst_bps, int K, uint64_t from,
                          int S, uint64_t *kmer, orc_result *res) {
    memset(res, 0, sizeof *res);
    return unsafe_shift_from(seq, src_bps, dst_bps, n_coding_elements(K, dst_bps), K, from, S, kmer, res);
}

/* ------------------------------------------------------------------------ */
/* src/transformations.jl:1-10 */
INL void kmer_reverse(const uint64_t *in, const int N, const int K, const int bps, uint64_t *out) {
    uint64_t tmp[ORC_MAX_N];
    for (int i = 0; i < N; ++i) tmp[i] = reversebits(in[N - 1 - i], bps); /* map(reversebits, reverse(x.data)) */
    rightshift_carry(tmp, N, bits_unused(K, bps), 0);
    for (int i = 0; i < N; ++i) out[i] = tmp[i];
}

/* src/transformations.jl:14-25 */
INL void kmer_complement(const uint64_t *in, const int N, const int K, const int bps, uint64_t *out) {
    if (K == 0) return;
    for (int i = 0; i < N; ++i) out[i] = complement_bitpar(in[i], bps);
    if (bps == 2) out[0] &= get_mask(K, bps); /* only the 2-bit method masks (:24) */
}

/* src/transformations.jl:32-34 */
INL void kmer_reverse_complement(const uint64_t *in, const int N, const int K, const int bps, uint64_t *out) {
    uint64_t tmp[ORC_MAX_N];
    for (int i = 0; i < N; ++i) tmp[i] = in[i];
    kmer_complement(in, N, K, bps, tmp);
    kmer_reverse(tmp, N, K, bps, out);
}

/* src/kmer.jl:176-178: cmp(x.data, y.data) for equal K -- lexicographic, head first */
INL int kmer_cmp(const uint64_t *x, const uint64_t *y, const int N) {
    for (int i = 0; i < N; ++i) {
        if (x[i] < y[i]) return -1;
        if (x[i] > y[i]) return 1;
    }
    return 0;
}

/* src/kmer.jl:255-261 (FX_CONSTANT :218) */
INL uint64_t fx_hash(const uint64_t *data, const int N, uint64_t h) {
    for (int i = 0; i < N; ++i) {
        uint64_t rot = (h << 5) | (h >> 59); /* bitrotate(h, 5) */
        h = (rot ^ data[i]) * 0x517cc1b727220a95ull;
    }
    return h;
}

void orc_reverse(const uint64_t *in, int K, int bps, uint64_t *out) {
    kmer_reverse(in, n_coding_elements(K, bps), K, bps, out);
}
void orc_complement(const uint64_t *in, int K, int bps, uint64_t *out) {
    kmer_complement(in, n_coding_elements(K, bps), K, bps, out);
}
void orc_reverse_complement(const uint64_t *in, int K, int bps, uint64_t *out) {
    kmer_reverse_complement(in, n_coding_elements(K, bps), K, bps, out);
}
/* src/transformations.jl:36-39: ifelse(x < rc, x, rc) */
void orc_canonical_kmer(const uint64_t *in, int K, int bps, uint64_t *out) {
    int N = n_coding_elements(K, bps);
    uint64_t rc[ORC_MAX_N];
    kmer_reverse_complement(in, N, K, bps, rc);
    const uint64_t *pick = kmer_cmp(in, rc, N) == -1 ? in : rc;
    for (int i = 0; i < N; ++i) out[i] = pick[i];
}
/* src/transformations.jl:41: x <= reverse_complement(x) */
int orc_iscanonical(const uint64_t *in, int K, int bps) {
    int N = n_coding_elements(K, bps);
    uint64_t rc[ORC_MAX_N];
    kmer_reverse_complement(in, N, K, bps, rc);
    return kmer_cmp(in, rc, N) <= 0;
}
int orc_cmp(const uint64_t *x, const uint64_t *y, int N) { return kmer_cmp(x, y, N); }
uint64_t orc_fx_hash(const uint64_t *kmer, int N, uint64_t h) { return fx_hash(kmer, N, h); }

/* src/counting.jl:1-8: BioSequences._n_gc(x::Kmer{<:TwoBit}) = count(isGC, x) */
int orc_n_gc(const uint64_t *kmer, int N) {
    uint64_t mask = 0x5555555555555555ull;
    int n = 0;
    for (int i = 0; i < N; ++i) n += count_ones64((kmer[i] ^ (kmer[i] >> 1)) & mask);
    return n;
}

/* src/kmer.jl:305-326 -- value returned as (hi, lo) of a UInt128; the return
 * value is the width in bits of the Julia result type (8,16,32,64,128), or -1
 * for the ArgumentError branch. */
int orc_as_integer(const uint64_t *t, int K, int bps, uint64_t *hi, uint64_t *lo) {
    *hi = 0;
    *lo = 0;
    if (K == 0) return 8; /* isempty(x) && return 0x00 */
    int bits = K * bps;
    if (bits <= 8) { *lo = t[0] & 0xff; return 8; }
    if (bits <= 16) { *lo = t[0] & 0xffff; return 16; }
    if (bits <= 32) { *lo = t[0] & 0xffffffffull; return 32; }
    if (bits <= 64) { *lo = t[0]; return 64; }
    if (bits <= 128) { *hi = t[0]; *lo = t[1]; return 128; }
    return -1;
}

/* src/kmer.jl:361-384 */
int orc_from_integer(uint64_t hi, uint64_t lo, int K, int bps, uint64_t *out) {
    int bits = K * bps;
    if (bits == 0) return 0; /* zero_kmer(T) */
    if (bits > 128) return -1;
    if (bits <= 64) {
        out[0] = lo & get_mask(K, bps);
    } else {
        out[0] = hi & get_mask(K, bps);
        out[1] = lo;
    }
    return 0;
}

/* src/construction.jl:213-219 (build_kmer(::Copyable, T, s::LongSequence)) */
int orc_kmer_from_longseq(const uint64_t *seq, uint64_t len, int K, int bps, uint64_t *out) {
    if (len != (uint64_t)K) return ORC_E_BADARG; /* "Length of sequence must be K elements" */
    int N = n_coding_elements(K, bps);
    for (int i = 0; i < N; ++i) out[i] = reversebits(seq[i], bps);
    rightshift_carry(out, N, bits_unused(K, bps), 0);
    return 0;
}

/* src/construction.jl:289-324 (LongSequence{A}(kmer): _fill_unshift! / _fill_shift!) */
void orc_longseq_from_kmer(const uint64_t *kmer, int K, int bps, uint64_t *data) {
    if (K == 0) return;
    int nce = n_coding_elements(K, bps);
    int bu = bits_unused(K, bps);
    if (bu == 0) { /* _fill_unshift! :303-308 */
        for (int i = 0; i < nce; ++i) data[i] = reversebits(kmer[i], bps);
        return;
    }
    /* _fill_shift! :309-324 */
    uint64_t left_mask = ((uint64_t)1 << (64 - bu)) - 1;
    uint64_t right_mask = ~left_mask;
    int leftsh = bu, rightsh = 64 - bu;
    for (int i = 0; i < nce - 1; ++i) {
        uint64_t chunk = left_shift(kmer[i] & left_mask, leftsh);
        chunk |= right_shift(kmer[i + 1] & right_mask, rightsh);
        data[i] = reversebits(chunk, bps);
    }
    data[nce - 1] = reversebits(left_shift(kmer[nce - 1] & left_mask, leftsh), bps);
}

/* ------------------------------------------------------------------------ */
/* FwKmers: src/iterators/FwKmers.jl:40-43 (length), :57-66 (first), :88-115 (step) */
INL int fw_kmers_impl(const uint64_t *seq, uint64_t len, const int src_bps, const int dst_bps,
                      const int N, const int K, uint64_t *out, orc_result *res) {
    uint64_t kmer[ORC_MAX_N];
    if (len < (uint64_t)K) return 0;                                        /* :63 */
    if (unsafe_extract(seq, src_bps, dst_bps, N, K, 1, kmer, res)) return ORC_E_ENCODE; /* :64 */
    uint64_t i = (uint64_t)K + 1;                                           /* :65 */
    for (;;) {
        if (out)
            for (int w = 0; w < N; ++w) out[res->n_out * N + w] = kmer[w];
        res->n_out++;
        if (i > len) return 0;                                              /* :90,:98,:110 */
        uint64_t enc;
        if (fetch_recoded(seq, i, src_bps, dst_bps, &enc, res)) return ORC_E_ENCODE;
        shift_encoding(kmer, N, K, dst_bps, enc);                           /* :92,:100,:113 */
        ++i;
    }
}

/* FwRvIterator: src/iterators/CanonicalKmers.jl:61-66 (first), :94-144 (step);
 * CanonicalKmers :220-225 min-select; fx_hash src/kmer.jl:255-261.
 * mode bits: 1 = emit (fw, rv); 2 = emit canonical [+hash]; 4 = xor-reduce data[1] */
INL int fwrv_impl(const uint64_t *seq, uint64_t len, const int src_bps, const int dst_bps,
                  const int N, const int K, uint64_t *out_a, uint64_t *out_b, const int canonical,
                  uint64_t seed, uint64_t *xor_acc, orc_result *res) {
    uint64_t fw[ORC_MAX_N], rv[ORC_MAX_N];
    if (len < (uint64_t)K) return 0;                                        /* :62 */
    if (unsafe_extract(seq, src_bps, dst_bps, N, K, 1, fw, res)) return ORC_E_ENCODE; /* :63 */
    kmer_reverse_complement(fw, N, K, dst_bps, rv);                         /* :64 */
    uint64_t i = (uint64_t)K + 1;                                           /* :65 */
    for (;;) {
        if (canonical) {
            const uint64_t *c = kmer_cmp(fw, rv, N) == -1 ? fw : rv;        /* :224 fw < rv ? fw : rv */
            if (out_a)
                for (int w = 0; w < N; ++w) out_a[res->n_out * N + w] = c[w];
            if (out_b) out_b[res->n_out] = fx_hash(c, N, seed);
            if (xor_acc) *xor_acc ^= c[0];                                  /* test/benchmark.jl:12 */
        } else {
            if (out_a)
                for (int w = 0; w < N; ++w) out_a[res->n_out * N + w] = fw[w];
            if (out_b)
                for (int w = 0; w < N; ++w) out_b[res->n_out * N + w] = rv[w];
        }
        res->n_out++;
        if (i > len) return 0;                                              /* :100,:113,:124,:137 */
        uint64_t fenc, renc;
        if (IS_BYTES(src_bps)) {                     /* AsciiEncode: :146-174; GenericRecoding: :81-91 (complement(symbol)) */
            if (fetch_recoded(seq, i, src_bps, dst_bps, &fenc, res)) return ORC_E_ENCODE;
            renc = dst_bps == 4 ? complement_nibble(fenc) : (fenc ^ 0x03); /* :161-165 */
            shift_encoding(fw, N, K, dst_bps, fenc);
            shift_first_encoding(rv, N, K, dst_bps, renc);
            ++i;
            continue;
        }
        uint64_t encoding = extract_encoded_element(seq, i, src_bps);
        if (src_bps == 2 && dst_bps == 2) {          /* Copyable, TwoBit: :94-105 */
            fenc = encoding;
            renc = encoding ^ 0x03;
        } else if (src_bps == 4 && dst_bps == 4) {   /* Copyable, FourBit: :107-120 */
            fenc = encoding;
            renc = complement_nibble(encoding);
        } else if (src_bps == 2 && dst_bps == 4) {   /* TwoToFour: :122-129 */
            fenc = left_shift((uint64_t)1, (int64_t)encoding);
            renc = left_shift((uint64_t)1, (int64_t)(encoding ^ 0x03));
        } else {                                     /* FourToTwo: :131-144 */
            if (count_ones64(encoding) != 1) return throw_uncertain(res, i, encoding);
            fenc = (uint64_t)trailing_zeros64(encoding);
            renc = fenc ^ 0x03;
        }
        shift_encoding(fw, N, K, dst_bps, fenc);
        shift_first_encoding(rv, N, K, dst_bps, renc);
        ++i;
    }
}

/* UnambiguousKmers: src/iterators/UnambiguousKmers.jl:64-77 (Copyable -> FwKmers with
 * index i-K+1), :79-86 (initial state), :88-106 (any other scheme: symbol by symbol), :109-132 (text),
 * :134-148 (FourToTwo loop). dst is TwoBit. */
INL int unambiguous_impl(const uint64_t *seq, uint64_t len, const int src_bps, const int N,
                         const int K, uint64_t *out_kmers, int64_t *out_starts, orc_result *res) {
    uint64_t kmer[ORC_MAX_N];
    if (src_bps == 2) { /* :64-77 */
        if (len < (uint64_t)K) return 0;
        unsafe_extract(seq, 2, 2, N, K, 1, kmer, res);
        uint64_t i = (uint64_t)K + 1;
        int64_t start = 1; /* :68 */
        for (;;) {
            if (out_kmers)
                for (int w = 0; w < N; ++w) out_kmers[res->n_out * N + w] = kmer[w];
            if (out_starts) out_starts[res->n_out] = start;
            res->n_out++;
            if (i > len) return 0;
            shift_encoding(kmer, N, K, 2, extract_encoded_element(seq, i, 2));
            start = (int64_t)i - K + 1; /* :76 */
            ++i;
        }
    }
    /* :79-86: state = (zero kmer, K, 1) */
    for (int w = 0; w < N; ++w) kmer[w] = 0;
    int64_t remaining = K;
    uint64_t index = 1;
    if (src_bps == ORC_SRC_SYMBOLS) { /* :88-106, the method of every other RecodingScheme (GenericRecoding: one symbol per byte) */
        for (;;) {
            while (remaining != 0) {
                if (index > len) return 0;
                uint8_t symbol = ((const uint8_t *)seq)[index - 1]; /* convert(eltype(kmer), seq[index]): DNA <-> RNA keeps the value */
                index += 1;
                if (symbol < 16 && count_ones64(symbol) > 1) { /* isambiguous(symbol) */
                    remaining = K;
                } else {
                    /* shift(kmer, symbol) encodes the symbol in the 2-bit alphabet: the gap (and a byte that is no symbol) throws */
                    if (symbol == 0 || symbol > 15) return throw_uncertain(res, index - 1, symbol);
                    remaining -= 1;
                    shift_encoding(kmer, N, K, 2, (uint64_t)trailing_zeros64(symbol));
                }
            }
            if (out_kmers)
                for (int w = 0; w < N; ++w) out_kmers[res->n_out * N + w] = kmer[w];
            if (out_starts) out_starts[res->n_out] = (int64_t)index - K;
            res->n_out++;
            remaining = 1;
        }
    }
    if (IS_ASCII(src_bps)) { /* :109-132 with ASCII_SKIPPING_LUT */
        for (;;) {
            while (remaining != 0) {
                if (index > len) return 0;
                uint8_t byte = ((const uint8_t *)seq)[index - 1];
                index += 1;
                uint8_t encoding = ascii_skipping_lut(byte);
                if (encoding == 0xff) return throw_uncertain(res, index - 1, byte); /* :122-123 */
                if (encoding == 0xf0) {
                    remaining = K;
                } else {
                    remaining -= 1;
                    shift_encoding(kmer, N, K, 2, encoding);
                }
            }
            if (out_kmers)
                for (int w = 0; w < N; ++w) out_kmers[res->n_out * N + w] = kmer[w];
            if (out_starts) out_starts[res->n_out] = (int64_t)index - K;
            res->n_out++;
            remaining = 1;
        }
    }
    for (;;) {
        while (remaining != 0) { /* :140-146 */
            if (index > len) return 0;
            uint64_t encoding = extract_encoded_element(seq, index, 4);
            shift_encoding(kmer, N, K, 2, (uint64_t)trailing_zeros64(encoding)); /* tz(0)==64, flushed later */
            index += 1;
            remaining = count_ones64(encoding) == 1 ? remaining - 1 : K;
        }
        if (out_kmers)
            for (int w = 0; w < N; ++w) out_kmers[res->n_out * N + w] = kmer[w];
        if (out_starts) out_starts[res->n_out] = (int64_t)index - K; /* :147 */
        res->n_out++;
        remaining = 1; /* state (kmer, 1, index) */
    }
}

/* SpacedKmers: src/iterators/SpacedKmers.jl:38-42 (length), :92-105 (first), :121-139 (step) */
INL int spaced_impl(const uint64_t *seq, uint64_t len, const int src_bps, const int dst_bps,
                    const int N, const int K, const int J, uint64_t *out, orc_result *res) {
    uint64_t kmer[ORC_MAX_N];
    if (len < (uint64_t)K) return 0;                                         /* :96 */
    if (unsafe_extract(seq, src_bps, dst_bps, N, K, 1, kmer, res)) return ORC_E_ENCODE; /* :97-102 */
    uint64_t i = 1 + (uint64_t)(J > K ? J : K);                              /* :103 */
    uint64_t minkj = (uint64_t)(K < J ? K : J);
    for (;;) {
        if (out)
            for (int w = 0; w < N; ++w) out[res->n_out * N + w] = kmer[w];
        res->n_out++;
        if (i + minkj > len + 1) return 0; /* :130  i > lastindex - min(K,J) + 1 */
        if (J >= K) {                                                        /* :133-134 */
            if (unsafe_extract(seq, src_bps, dst_bps, N, K, i, kmer, res)) return ORC_E_ENCODE;
        } else {                                                             /* :136 */
            if (unsafe_shift_from(seq, src_bps, dst_bps, N, K, i, J, kmer, res)) return ORC_E_ENCODE;
        }
        i += (uint64_t)J;                                                    /* :131 */
    }
}

/* Minimizers: the example the reference publishes on top of its public primitives
 * (docs/src/replacements.md:33-51, test/benchmark.jl:96-110):
 *     kmer = unsafe_extract(R, T, seq, i); hash = fx_hash(kmer)
 *     for offset in 0:W-2
 *         new_kmer = unsafe_shift_from(R, kmer, seq, i+K+offset, Val(1)); new_hash = fx_hash(new_kmer)
 *         if new_hash < hash; hash = new_hash; kmer = new_kmer; end
 * mode 0 restates it literally -- note that the symbol is shifted into `kmer`, the CURRENT MINIMUM,
 * not into the previous window position.  mode 1 is the true sliding-window minimizer (the kmer
 * with the smallest fx_hash among the W consecutive kmers, leftmost on ties).
 * Windows start at i = 1, 1+stride, ... while the window fits (i + K + W - 2 <= len). */
static int check_args(int src_bps, int dst_bps, int K, orc_result *res);

int orc_minimizers(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K, int W, int stride,
                   int mode, uint64_t *out_kmers, orc_result *res) {
    if (check_args(src_bps, dst_bps, K, res)) return ORC_E_BADARG;
    if (W < 1 || stride < 1) { res->status = ORC_E_BADARG; return ORC_E_BADARG; }
    const int N = n_coding_elements(K, dst_bps);
    const uint64_t span = (uint64_t)K + (uint64_t)W - 1;
    if (len < span) return 0;
    for (uint64_t i = 1; i + span - 1 <= len; i += (uint64_t)stride) {
        uint64_t kmer[ORC_MAX_N], cur[ORC_MAX_N], nk[ORC_MAX_N];
        if (unsafe_extract(seq, src_bps, dst_bps, N, K, i, kmer, res)) return ORC_E_ENCODE;
        for (int w = 0; w < N; ++w) cur[w] = kmer[w];
        uint64_t hash = fx_hash(kmer, N, 0);
        for (int offset = 0; offset <= W - 2; ++offset) {
            const uint64_t *from = mode == 0 ? kmer : cur;
            for (int w = 0; w < N; ++w) nk[w] = from[w];
            if (unsafe_shift_from(seq, src_bps, dst_bps, N, K, i + (uint64_t)K + (uint64_t)offset, 1, nk, res))
                return ORC_E_ENCODE;
            for (int w = 0; w < N; ++w) cur[w] = nk[w];
            uint64_t new_hash = fx_hash(nk, N, 0);
            if (new_hash < hash) {
                hash = new_hash;
                for (int w = 0; w < N; ++w) kmer[w] = nk[w];
            }
        }
        if (out_kmers)
            for (int w = 0; w < N; ++w) out_kmers[res->n_out * N + w] = kmer[w];
        res->n_out++;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* public wrappers: constant-fold the common geometries so that the cpu_baseline
 * timing is of specialised code (Julia specialises on A, K, N at compile time). */
static int check_args(int src_bps, int dst_bps, int K, orc_result *res) {
    memset(res, 0, sizeof *res);
    if ((src_bps != 2 && src_bps != 4 && !IS_BYTES(src_bps)) || (dst_bps != 2 && dst_bps != 4) || K < 1 ||
        n_coding_elements(K, dst_bps) > ORC_MAX_N) {
        res->status = ORC_E_BADARG; /* FwKmers.jl:31-35 "K must be at least 1" */
        return ORC_E_BADARG;
    }
    return 0;
}

#define DISPATCH(CALL_WITH)                                         \
    do {                                                            \
        if (src_bps == 4 && dst_bps == 2 && N == 1) { CALL_WITH(4, 2, 1); } \
        else if (src_bps == 2 && dst_bps == 2 && N == 1) { CALL_WITH(2, 2, 1); } \
        else if (src_bps == 4 && dst_bps == 2 && N == 2) { CALL_WITH(4, 2, 2); } \
        else if (src_bps == 2 && dst_bps == 2 && N == 2) { CALL_WITH(2, 2, 2); } \
        else { CALL_WITH(src_bps, dst_bps, N); }                    \
    } while (0)

int orc_fw_kmers(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K,
                 uint64_t *out, orc_result *res) {
    if (check_args(src_bps, dst_bps, K, res)) return ORC_E_BADARG;
    int N = n_coding_elements(K, dst_bps);
#define CALL_FW(S, D, NN) return fw_kmers_impl(seq, len, S, D, NN, K, out, res)
    DISPATCH(CALL_FW);
#undef CALL_FW
}

int orc_fwrv(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K,
             uint64_t *out_fw, uint64_t *out_rv, orc_result *res) {
    if (check_args(src_bps, dst_bps, K, res)) return ORC_E_BADARG;
    int N = n_coding_elements(K, dst_bps);
#define CALL_FWRV(S, D, NN) return fwrv_impl(seq, len, S, D, NN, K, out_fw, out_rv, 0, 0, 0, res)
    DISPATCH(CALL_FWRV);
#undef CALL_FWRV
}

int orc_canonical(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K,
                  uint64_t *out_kmers, uint64_t *out_hashes, uint64_t seed, orc_result *res) {
    if (check_args(src_bps, dst_bps, K, res)) return ORC_E_BADARG;
    int N = n_coding_elements(K, dst_bps);
#define CALL_CAN(S, D, NN) return fwrv_impl(seq, len, S, D, NN, K, out_kmers, out_hashes, 1, seed, 0, res)
    DISPATCH(CALL_CAN);
#undef CALL_CAN
}

uint64_t orc_reduce_xor_canonical(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K,
                                  orc_result *res) {
    uint64_t acc = 0;
    if (check_args(src_bps, dst_bps, K, res)) return 0;
    int N = n_coding_elements(K, dst_bps);
#define CALL_RED(S, D, NN) do { fwrv_impl(seq, len, S, D, NN, K, 0, 0, 1, 0, &acc, res); return acc; } while (0)
    DISPATCH(CALL_RED);
#undef CALL_RED
}

int orc_unambiguous(const uint64_t *seq, uint64_t len, int src_bps, int K,
                    uint64_t *out_kmers, int64_t *out_starts, orc_result *res) {
    if (check_args(src_bps, 2, K, res)) return ORC_E_BADARG;
    int N = n_coding_elements(K, 2);
    if (src_bps == 4 && N == 1) return unambiguous_impl(seq, len, 4, 1, K, out_kmers, out_starts, res);
    return unambiguous_impl(seq, len, src_bps, N, K, out_kmers, out_starts, res);
}

int orc_spaced(const uint64_t *seq, uint64_t len, int src_bps, int dst_bps, int K, int J,
               uint64_t *out, orc_result *res) {
    if (check_args(src_bps, dst_bps, K, res)) return ORC_E_BADARG;
    if (J < 1) { res->status = ORC_E_BADARG; return ORC_E_BADARG; } /* SpacedKmers.jl:29-30 */
    int N = n_coding_elements(K, dst_bps);
    if (src_bps == 4 && dst_bps == 2 && N == 1) return spaced_impl(seq, len, 4, 2, 1, K, J, out, res);
    return spaced_impl(seq, len, src_bps, dst_bps, N, K, J, out, res);
}

/* ------------------------------------------------------------------------ */
/* Synthetic input generator (the build's own).  Counter based
 * so that any word of any shard can be produced independently:
 *   rand64(seed, idx) = SplitMix64 finaliser of seed + (idx+1)*golden
 *   base b (0-based) has 2-bit code (rand64(seed, b/32) >> 2*(b%32)) & 3
 *   2-bit word w  = rand64(seed, w)                (32 bases)
 *   4-bit word w  = one-hot nibbles of bases 16w..16w+15
 *   optional ambiguity: base b becomes N (0b1111) when the 16-bit lane
 *   (rand64(seed ^ 0xA5A5.., b/4) >> 16*(b%4)) & 0xffff < ambig_per_65536
 *   (4-bit only; mirrors test/utils.jl:22-24 p(N)=0.04 when set to 2621). */
uint64_t orc_synth_rand64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_synth_words(uint64_t seed, uint64_t first_word, uint64_t n_words, int bps,
                     uint32_t ambig_per_65536, uint64_t *out) {
    for (uint64_t k = 0; k < n_words; ++k) {
        uint64_t w = first_word + k;
        if (bps == 2) {
            out[k] = orc_synth_rand64(seed, w);
            continue;
        }
        uint64_t r = orc_synth_rand64(seed, w >> 1) >> (32 * (w & 1));
        uint64_t word = 0;
        for (int j = 0; j < 16; ++j) {
            uint64_t nib = (uint64_t)1 << ((r >> (2 * j)) & 3);
            if (ambig_per_65536) {
                uint64_t b = w * 16 + (uint64_t)j;
                uint64_t u = (orc_synth_rand64(seed ^ 0xA5A5A5A5A5A5A5A5ull, b >> 2) >> (16 * (b & 3))) & 0xffff;
                if (u < ambig_per_65536) nib = 0xF;
            }
            word |= nib << (4 * j);
        }
        out[k] = word;
    }
}
