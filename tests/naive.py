"""Independent naive checker ("slice each window from text, pack it").

Shares no code with oracle/kmers_oracle.c or the HIP kernels: kmers are built
with Python big-int arithmetic straight from the layout rule of
src/kmer.jl:33-40 (first symbol in the most significant used bits, unused bits
= top bits of word 1, zero), LongSequence words from the little-endian rule of
SURVEY.md section 2.1.  This is the third leg of the reference's own test style
(test/runtests.jl:686-688, :740-760, :782-786).
"""
import numpy as np

MASK64 = (1 << 64) - 1
FX_CONSTANT = 0x517CC1B727220A95  # src/kmer.jl:218

# BioSymbols encodings (absent dependency, published tables)
DNA4 = {"-": 0, "A": 1, "C": 2, "M": 3, "G": 4, "R": 5, "S": 6, "V": 7, "T": 8, "W": 9,
        "Y": 10, "H": 11, "K": 12, "D": 13, "B": 14, "N": 15, "U": 8}
DNA4_INV = {v: k for k, v in DNA4.items() if k != "U"}
DNA2 = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}
AA = {c: i for i, c in enumerate("ARNDCQEGHILKMFPSTWYVOUBJZX*-")}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "U": "A", "M": "K", "K": "M", "R": "Y", "Y": "R",
        "W": "W", "S": "S", "V": "B", "B": "V", "H": "D", "D": "H", "N": "N", "-": "-"}


def table(bps_or_name):
    if bps_or_name in (2, "dna2", "rna2"):
        return DNA2, 2
    if bps_or_name in (4, "dna4", "rna4"):
        return DNA4, 4
    if bps_or_name in (8, "aa"):
        return AA, 8
    raise ValueError(bps_or_name)


def n_words(K, bps):
    return (K * bps + 63) // 64


def kmer_words(text, alphabet):
    """Kmer{A,K,N}.data for the symbols of `text` (src/kmer.jl:33-40)."""
    tab, bps = table(alphabet)
    v = 0
    for ch in text.upper():
        v = (v << bps) | tab[ch]
    N = n_words(len(text), bps)
    return tuple((v >> (64 * (N - 1 - i))) & MASK64 for i in range(N))


def kmer_text(words, K, alphabet):
    tab, bps = table(alphabet)
    inv = {v: k for k, v in tab.items() if k != "U"}
    v = 0
    for w in words:
        v = (v << 64) | int(w)
    return "".join(inv[(v >> (bps * (K - 1 - t))) & ((1 << bps) - 1)] for t in range(K))


def longseq_words(text, alphabet):
    """LongSequence.data (little-endian by symbol) as a numpy uint64 array (+1 pad word)."""
    tab, bps = table(alphabet)
    n = (len(text) * bps + 63) // 64
    out = [0] * (n + 1)
    for i, ch in enumerate(text.upper()):
        bit = i * bps
        out[bit >> 6] |= tab[ch] << (bit & 63)
    return np.array(out, dtype=np.uint64)


def ascii_words(text):
    """A String / byte source as the oracle and the C ABI take it: the bytes, zero padded to whole
    8-byte words (+1 spare), viewed as uint64 (byte i = bits 8i.. of word i // 8)."""
    raw = text.encode("latin-1") if isinstance(text, str) else bytes(text)
    raw = raw + b"\0" * ((-len(raw)) % 8 + 8)
    return np.frombuffer(raw, dtype=np.uint64).copy()


def revcomp_text(text):
    return "".join(COMP[c] for c in reversed(text.upper()))


def is_certain(text):
    return all(c in "ACGTU" for c in text.upper())


def fx_hash(words, h=0):
    for w in words:
        rot = ((h << 5) | (h >> 59)) & MASK64
        h = ((rot ^ int(w)) * FX_CONSTANT) & MASK64
    return h


def fw_kmers(text, K, dst):
    return [kmer_words(text[i:i + K], dst) for i in range(0, max(0, len(text) - K + 1))]


def fwrv(text, K, dst):
    return [(kmer_words(text[i:i + K], dst), kmer_words(revcomp_text(text[i:i + K]), dst))
            for i in range(0, max(0, len(text) - K + 1))]


def canonical(text, K, dst):
    return [min(f, r) for f, r in fwrv(text, K, dst)]


def unambiguous(text, K):
    return [(kmer_words(text[i:i + K], 2), i + 1)
            for i in range(0, max(0, len(text) - K + 1)) if is_certain(text[i:i + K])]


def spaced(text, K, J, dst):
    return [kmer_words(text[i:i + K], dst) for i in range(0, max(0, len(text) - K + 1), J)] \
        if len(text) >= K else []


def first_ambiguous(text, positions):
    """1-based position of the first non-ACGT symbol among the inspected positions."""
    for p in positions:
        if text[p - 1].upper() not in "ACGTU":
            return p
    return None


def random_text(rng, n, p_amb=0.0):
    letters = np.array(list("ACGT"))
    t = letters[rng.integers(0, 4, size=n)]
    if p_amb > 0:
        amb = np.array(list("NMRWSYKVHDB-"))
        m = rng.random(n) < p_amb
        t[m] = amb[rng.integers(0, len(amb), size=int(m.sum()))]
    return "".join(t.tolist())
