"""ctypes loader for the CPU oracle (oracle/kmers_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORC_MAX_N = 64
OK, E_ENCODE, E_BADARG = 0, 1, 2


class Result(C.Structure):
    _fields_ = [("n_out", C.c_uint64), ("status", C.c_int32), ("err_enc", C.c_uint32),
                ("err_pos", C.c_uint64)]


def build(native=False, out_dir=None):
    """Compile the oracle with gcc. native=True adds -march=native (cpu_baseline timing on
    the box it runs on); the default build is portable so the .so can travel."""
    out_dir = out_dir or _HERE
    name = "libkmers_oracle_native.so" if native else "libkmers_oracle.so"
    out = os.path.join(out_dir, name)
    src = os.path.join(_HERE, "kmers_oracle.c")
    hdr = os.path.join(_HERE, "kmers_oracle.h")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(hdr)):
        return out
    march = "-march=native" if native else "-march=x86-64-v2"
    tmp = f"{out}.{os.getpid()}.tmp"  # atomic: several ranks may call this at once
    cmd = ["gcc", "-O3", "-std=c11", "-fPIC", "-Wall", "-Wextra", "-Wno-return-type", march,
           "-shared", "-o", tmp, src]
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)
    return out


_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)


def _ptr(a, typ=_u64p):
    return None if a is None else a.ctypes.data_as(typ)


class Oracle:
    def __init__(self, path=None, native=False):
        path = path or build(native=native)
        L = self.lib = C.CDLL(path)
        R = C.POINTER(Result)
        L.orc_n_coding_elements.restype = C.c_int
        L.orc_bits_unused.restype = C.c_int
        L.orc_elements_in_head.restype = C.c_int
        L.orc_get_mask.restype = C.c_uint64
        L.orc_fx_hash.restype = C.c_uint64
        L.orc_fx_hash.argtypes = [_u64p, C.c_int, C.c_uint64]
        L.orc_fw_kmers.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p, R]
        L.orc_fwrv.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p, _u64p, R]
        L.orc_canonical.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, _u64p, _u64p,
                                    C.c_uint64, R]
        L.orc_unambiguous.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, _u64p, _i64p, R]
        L.orc_spaced.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _u64p, R]
        L.orc_reduce_xor_canonical.restype = C.c_uint64
        L.orc_reduce_xor_canonical.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, R]
        L.orc_unsafe_extract.argtypes = [_u64p, C.c_int, C.c_int, C.c_int, C.c_uint64, _u64p, R]
        L.orc_unsafe_shift_from.argtypes = [_u64p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int,
                                            _u64p, R]
        L.orc_shift_encoding.argtypes = [_u64p, C.c_int, C.c_int, C.c_uint64]
        L.orc_shift_first_encoding.argtypes = [_u64p, C.c_int, C.c_int, C.c_uint64]
        for f in (L.orc_reverse, L.orc_complement, L.orc_reverse_complement, L.orc_canonical_kmer):
            f.argtypes = [_u64p, C.c_int, C.c_int, _u64p]
        L.orc_iscanonical.argtypes = [_u64p, C.c_int, C.c_int]
        L.orc_cmp.argtypes = [_u64p, _u64p, C.c_int]
        L.orc_as_integer.argtypes = [_u64p, C.c_int, C.c_int, _u64p, _u64p]
        L.orc_from_integer.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, _u64p]
        L.orc_kmer_from_longseq.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, _u64p]
        L.orc_longseq_from_kmer.argtypes = [_u64p, C.c_int, C.c_int, _u64p]
        L.orc_n_gc.argtypes = [_u64p, C.c_int]
        L.orc_minimizers.argtypes = [_u64p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u64p, R]
        L.orc_synth_rand64.restype = C.c_uint64
        L.orc_synth_rand64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_synth_words.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, _u64p]
        L.orc_leftshift_carry.restype = C.c_uint64
        L.orc_leftshift_carry.argtypes = [_u64p, C.c_int, C.c_int64, C.c_uint64]
        L.orc_rightshift_carry.restype = C.c_uint64
        L.orc_rightshift_carry.argtypes = [_u64p, C.c_int, C.c_int64, C.c_uint64]

    # ---- geometry ----
    def nwords(self, K, bps):
        return self.lib.orc_n_coding_elements(K, bps)

    # ---- helpers ----
    @staticmethod
    def _seq(words):
        a = np.ascontiguousarray(words, dtype=np.uint64)
        if a.size == 0:
            a = np.zeros(1, dtype=np.uint64)
        return a

    def _kw(self, words):
        a = np.zeros(ORC_MAX_N, dtype=np.uint64)
        a[:len(words)] = np.array([int(w) for w in words], dtype=np.uint64)
        return a

    # ---- iterators ----
    def fw_kmers(self, seq, length, src_bps, dst_bps, K):
        seq = self._seq(seq)
        N = self.nwords(K, dst_bps)
        n = max(0, length - K + 1)
        out = np.zeros((n, N), dtype=np.uint64)
        res = Result()
        self.lib.orc_fw_kmers(_ptr(seq), length, src_bps, dst_bps, K, _ptr(out), C.byref(res))
        return out[:res.n_out], res

    def fwrv(self, seq, length, src_bps, dst_bps, K):
        seq = self._seq(seq)
        N = self.nwords(K, dst_bps)
        n = max(0, length - K + 1)
        fw = np.zeros((n, N), dtype=np.uint64)
        rv = np.zeros((n, N), dtype=np.uint64)
        res = Result()
        self.lib.orc_fwrv(_ptr(seq), length, src_bps, dst_bps, K, _ptr(fw), _ptr(rv), C.byref(res))
        return fw[:res.n_out], rv[:res.n_out], res

    def canonical(self, seq, length, src_bps, dst_bps, K, seed=0, hashes=True, out=None, out_h=None):
        seq = self._seq(seq)
        N = self.nwords(K, dst_bps)
        n = max(0, length - K + 1)
        km = out if out is not None else np.zeros((n, N), dtype=np.uint64)
        hs = out_h if out_h is not None else (np.zeros(n, dtype=np.uint64) if hashes else None)
        res = Result()
        self.lib.orc_canonical(_ptr(seq), length, src_bps, dst_bps, K, _ptr(km), _ptr(hs), seed,
                               C.byref(res))
        if hs is None:
            return km[:res.n_out], None, res
        return km[:res.n_out], hs[:res.n_out], res

    def unambiguous(self, seq, length, src_bps, K):
        seq = self._seq(seq)
        N = self.nwords(K, 2)
        n = max(0, length - K + 1)
        km = np.zeros((n, N), dtype=np.uint64)
        st = np.zeros(n, dtype=np.int64)
        res = Result()
        self.lib.orc_unambiguous(_ptr(seq), length, src_bps, K, _ptr(km), _ptr(st, _i64p), C.byref(res))
        return km[:res.n_out], st[:res.n_out], res

    def spaced(self, seq, length, src_bps, dst_bps, K, J):
        seq = self._seq(seq)
        N = self.nwords(K, dst_bps)
        n = 0 if length < K else (length - K) // J + 1
        out = np.zeros((n, N), dtype=np.uint64)
        res = Result()
        self.lib.orc_spaced(_ptr(seq), length, src_bps, dst_bps, K, J, _ptr(out), C.byref(res))
        return out[:res.n_out], res

    def minimizers(self, seq, length, src_bps, dst_bps, K, W, stride, mode=0):
        seq = self._seq(seq)
        N = self.nwords(K, dst_bps)
        span = K + W - 1
        n = 0 if length < span else (length - span) // stride + 1
        out = np.zeros((n, N), dtype=np.uint64)
        res = Result()
        self.lib.orc_minimizers(_ptr(seq), length, src_bps, dst_bps, K, W, stride, mode, _ptr(out), C.byref(res))
        return out[:res.n_out], res

    def reduce_xor_canonical(self, seq, length, src_bps, dst_bps, K):
        seq = self._seq(seq)
        res = Result()
        v = self.lib.orc_reduce_xor_canonical(_ptr(seq), length, src_bps, dst_bps, K, C.byref(res))
        return v, res

    # ---- single-kmer ops (kmer = tuple of ints, head first) ----
    def _unary(self, fn, words, K, bps):
        a = self._kw(words)
        o = np.zeros(ORC_MAX_N, dtype=np.uint64)
        fn(_ptr(a), K, bps, _ptr(o))
        return tuple(int(x) for x in o[:self.nwords(K, bps)])

    def reverse(self, w, K, bps):
        return self._unary(self.lib.orc_reverse, w, K, bps)

    def complement(self, w, K, bps):
        return self._unary(self.lib.orc_complement, w, K, bps)

    def reverse_complement(self, w, K, bps):
        return self._unary(self.lib.orc_reverse_complement, w, K, bps)

    def canonical_kmer(self, w, K, bps):
        return self._unary(self.lib.orc_canonical_kmer, w, K, bps)

    def iscanonical(self, w, K, bps):
        return bool(self.lib.orc_iscanonical(_ptr(self._kw(w)), K, bps))

    def fx_hash(self, w, seed=0):
        return int(self.lib.orc_fx_hash(_ptr(self._kw(w)), len(w), seed))

    def n_gc(self, w):
        return int(self.lib.orc_n_gc(_ptr(self._kw(w)), len(w)))

    def shift_encoding(self, w, K, bps, enc):
        a = self._kw(w)
        self.lib.orc_shift_encoding(_ptr(a), K, bps, enc)
        return tuple(int(x) for x in a[:self.nwords(K, bps)])

    def shift_first_encoding(self, w, K, bps, enc):
        a = self._kw(w)
        self.lib.orc_shift_first_encoding(_ptr(a), K, bps, enc)
        return tuple(int(x) for x in a[:self.nwords(K, bps)])

    def unsafe_extract(self, seq, src_bps, dst_bps, K, frm):
        seq = self._seq(seq)
        o = np.zeros(ORC_MAX_N, dtype=np.uint64)
        res = Result()
        self.lib.orc_unsafe_extract(_ptr(seq), src_bps, dst_bps, K, frm, _ptr(o), C.byref(res))
        return tuple(int(x) for x in o[:self.nwords(K, dst_bps)]), res

    def unsafe_shift_from(self, seq, src_bps, dst_bps, K, frm, S, kmer):
        seq = self._seq(seq)
        a = self._kw(kmer)
        res = Result()
        self.lib.orc_unsafe_shift_from(_ptr(seq), src_bps, dst_bps, K, frm, S, _ptr(a), C.byref(res))
        return tuple(int(x) for x in a[:self.nwords(K, dst_bps)]), res

    def as_integer(self, w, K, bps):
        hi, lo = C.c_uint64(), C.c_uint64()
        bits = self.lib.orc_as_integer(_ptr(self._kw(w)), K, bps, C.byref(hi), C.byref(lo))
        return (hi.value << 64) | lo.value, bits

    def from_integer(self, value, K, bps):
        o = np.zeros(ORC_MAX_N, dtype=np.uint64)
        rc = self.lib.orc_from_integer((value >> 64) & (2**64 - 1), value & (2**64 - 1), K, bps, _ptr(o))
        assert rc == 0
        return tuple(int(x) for x in o[:self.nwords(K, bps)])

    def kmer_from_longseq(self, seq, length, K, bps):
        seq = self._seq(seq)
        o = np.zeros(ORC_MAX_N, dtype=np.uint64)
        rc = self.lib.orc_kmer_from_longseq(_ptr(seq), length, K, bps, _ptr(o))
        assert rc == 0
        return tuple(int(x) for x in o[:self.nwords(K, bps)])

    def longseq_from_kmer(self, w, K, bps):
        o = np.zeros(ORC_MAX_N, dtype=np.uint64)
        self.lib.orc_longseq_from_kmer(_ptr(self._kw(w)), K, bps, _ptr(o))
        return o[:self.nwords(K, bps)].copy()

    # ---- synthetic data ----
    def synth_words(self, seed, first_word, n_words, bps, ambig_per_65536=0):
        out = np.zeros(max(1, n_words), dtype=np.uint64)
        self.lib.orc_synth_words(seed, first_word, n_words, bps, ambig_per_65536, _ptr(out))
        return out[:n_words]


_default = None


def get(native=False):
    global _default
    if native:
        return Oracle(native=True)
    if _default is None:
        _default = Oracle()
    return _default
