"""Host-side mirror of the reference's iterator / fx_hash / canonical API over the C ABI.

The reference is Julia and no Julia toolchain exists in this image, so the drop-in surface is
mirrored here in Python with the reference's names and argument meaning:

    Julia                                   here
    CanonicalDNAMers{31}(seq)               CanonicalDNAMers[31](seq)
    FwKmers{DNAAlphabet{2}, 21}(seq)        FwKmers[DNAAlphabet[2], 21](seq)
    SpacedDNAMers{21, 3}(seq)               SpacedDNAMers[21, 3](seq)
    collect(it), length(it), for x in it    collect(it), len(it), for x in it
    fx_hash(kmer, h), canonical(kmer), ...  same names

Every computation goes through libkmers_hip.so (kmers_jl_amd._capi); nothing is computed on the
CPU except packing text into LongSequence words and unpacking kmers back to text.
(The Julia `@ccall` shim with the same mapping is julia/KmersHIP.jl, see INTEGRATION.md.)
"""
import ctypes as C
import os

import numpy as np

from . import _capi

MASK64 = (1 << 64) - 1


# --------------------------------------------------------------------------------------------
# errors
class KmersError(RuntimeError):
    pass


class EncodeError(KmersError):
    """BioSequences.EncodeError (src/construction.jl:108-110)."""

    def __init__(self, alphabet, symbol, position, encoding):
        self.alphabet, self.symbol, self.position, self.encoding = alphabet, symbol, position, encoding
        super().__init__(f"cannot encode {symbol} in {alphabet}")


class UnsupportedError(KmersError):
    pass


# --------------------------------------------------------------------------------------------
# alphabets (BioSequences names; only what the hot path needs)
class _Alphabet:
    def __init__(self, kind, bits):
        self.kind, self.bits = kind, bits

    def __repr__(self):
        return f"{self.kind}Alphabet{{{self.bits}}}"

    def __eq__(self, other):
        return isinstance(other, _Alphabet) and (self.kind, self.bits) == (other.kind, other.bits)

    def __hash__(self):
        return hash((self.kind, self.bits))


class _AlphabetFamily:
    def __init__(self, kind):
        self.kind = kind

    def __getitem__(self, bits):
        if bits not in (2, 4):
            raise KmersError("nucleotide alphabets have 2 or 4 bits per symbol")
        return _Alphabet(self.kind, bits)


DNAAlphabet = _AlphabetFamily("DNA")
RNAAlphabet = _AlphabetFamily("RNA")

_ENC4 = {"-": 0, "A": 1, "C": 2, "M": 3, "G": 4, "R": 5, "S": 6, "V": 7, "T": 8, "W": 9, "Y": 10,
         "H": 11, "K": 12, "D": 13, "B": 14, "N": 15, "U": 8}
_ENC2 = {"A": 0, "C": 1, "G": 2, "T": 3, "U": 3}


def _decode_table(alphabet):
    t = "U" if alphabet.kind == "RNA" else "T"
    if alphabet.bits == 2:
        return ["A", "C", "G", t]
    inv = [None] * 16
    for ch, v in _ENC4.items():
        if ch not in "TU":
            inv[v] = ch
    inv[8] = t
    return inv


# --------------------------------------------------------------------------------------------
# context
class Context:
    """One kmers_ctx (one HIP stream).  `stream` may be a raw hipStream_t handle to borrow."""

    def __init__(self, device=0, stream=None):
        self.lib = _capi.load()
        h = C.c_void_p()
        rc = self.lib.kmers_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != _capi.OK:
            raise KmersError(
                f"kmers_ctx_create(device={device}) failed with {_capi.STATUS_NAMES.get(rc, rc)}: "
                "no usable MI355X/HIP device (this library has no CPU fallback)")
        self.handle = h
        self.device = device
        # test hook: run everything with a non-default workgroup tile (kmers per tile, a multiple of 512)
        tile = os.environ.get("KMERS_TILE_KMERS")
        if tile:
            self.set_param(_capi.PARAM_TILE_KMERS, int(tile))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.kmers_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self.lib.kmers_last_error(self.handle).decode()

    def check(self, rc, what):
        if rc not in (_capi.OK, _capi.E_ENCODE):
            msg = f"{what}: {_capi.STATUS_NAMES.get(rc, rc)}: {self.last_error()}"
            if rc == _capi.E_UNSUPPORTED:
                raise UnsupportedError(msg)
            raise KmersError(msg)
        return rc

    def set_param(self, param, value):
        self.check(self.lib.kmers_ctx_set_param(self.handle, param, value), "kmers_ctx_set_param")

    # device memory
    def alloc(self, nbytes, lone_output=False):
        """kmers_dev_alloc_role: `lone_output` = the only output array of the launches that fill it (its two halves are put into
        different region classes of HBM and such a launch writes it through two windows, include/kmers_hip.h)."""
        p = C.c_void_p()
        role = _capi.ALLOC_LONE_OUTPUT if lone_output else _capi.ALLOC_DEFAULT
        self.check(self.lib.kmers_dev_alloc_role(self.handle, nbytes, role, C.byref(p)), "kmers_dev_alloc_role")
        return p.value

    def free(self, ptr):
        if ptr and self.handle:
            self.lib.kmers_dev_free(self.handle, C.c_void_p(ptr))

    def last_launch_shape(self):
        """(threads per workgroup, kmers per tile, two write windows) of the most recent tile-kernel launch (kmers_last_launch_shape)."""
        t, k, sp = C.c_int(), C.c_int(), C.c_int()
        self.check(self.lib.kmers_last_launch_shape(self.handle, C.byref(t), C.byref(k), C.byref(sp)), "kmers_last_launch_shape")
        return t.value, k.value, sp.value

    # the device's class pool (include/kmers_hip.h): where alloc() of 128 MiB or more comes from
    def pool_info(self):
        """dict(held, in_use, n_classes, class_bytes, two_class_gbps, one_class_gbps) of the device's class pool (kmers_pool_info)."""
        h, u, n = C.c_size_t(), C.c_size_t(), C.c_int()
        cb = (C.c_size_t * _capi.POOL_CLASSES)()
        two, one = C.c_double(), C.c_double()
        self.check(self.lib.kmers_pool_info(self.handle, C.byref(h), C.byref(u), C.byref(n), cb, C.byref(two), C.byref(one)), "kmers_pool_info")
        return {"held": h.value, "in_use": u.value, "n_classes": n.value, "class_bytes": list(cb), "two_class_gbps": two.value, "one_class_gbps": one.value}

    POOL_STAT_NAMES = ("held", "in_use", "cached", "free", "cache_hits", "cache_misses", "evictions", "chunks_created", "chunks_returned",
                       "probes", "cached_blocks", "blocks_out")

    def pool_stats(self):
        """The pool's counters by name (kmers_pool_stats)."""
        v = (C.c_uint64 * _capi.POOL_STATS)()
        self.check(self.lib.kmers_pool_stats(self.handle, v, _capi.POOL_STATS), "kmers_pool_stats")
        return dict(zip(self.POOL_STAT_NAMES, list(v)))

    def pool_trim(self):
        r = C.c_size_t()
        self.check(self.lib.kmers_pool_trim(self.handle, C.byref(r)), "kmers_pool_trim")
        return r.value

    def pool_layout(self, ptr):
        """(chunk bytes, [region class of every 1 GiB handle]) of the pool block that holds `ptr`; an empty list if it is not one."""
        g, n = C.c_size_t(), C.c_size_t()
        self.check(self.lib.kmers_pool_layout(self.handle, C.c_void_p(ptr), C.byref(g), None, 0, C.byref(n)), "kmers_pool_layout")
        buf = (C.c_ubyte * max(n.value, 1))()
        self.check(self.lib.kmers_pool_layout(self.handle, C.c_void_p(ptr), C.byref(g), buf, n.value, C.byref(n)), "kmers_pool_layout")
        return g.value, list(buf[:n.value])

    def pool_selftest(self):
        """kmers_pool_selftest: raises on failure; returns whether a stale translation was seen without the pool's flush."""
        st = C.c_int()
        self.check(self.lib.kmers_pool_selftest(self.handle, C.byref(st)), "kmers_pool_selftest")
        return bool(st.value)

    def placement_probe(self, ptr_a, ptr_b, nbytes):
        """GB/s of two store streams side by side into two (still empty) device buffers; DESTRUCTIVE (kmers_placement_probe)."""
        g = C.c_double()
        self.check(self.lib.kmers_placement_probe(self.handle, C.c_void_p(ptr_a), C.c_void_p(ptr_b), nbytes, C.byref(g)), "kmers_placement_probe")
        return g.value

    def last_batch_pieces(self):
        """Pieces the most recent kmers_minhash_batch brought its pool up in (kmers_last_batch_pieces)."""
        n = C.c_uint64()
        self.check(self.lib.kmers_last_batch_pieces(self.handle, C.byref(n)), "kmers_last_batch_pieces")
        return n.value

    def host_alloc(self, nbytes):
        """Page-locked host memory (kmers_host_alloc): what an enqueued copy needs in order not to block its caller."""
        p = C.c_void_p()
        self.check(self.lib.kmers_host_alloc(self.handle, nbytes, C.byref(p)), "kmers_host_alloc")
        return p.value

    def host_free(self, ptr):
        if ptr and self.handle:
            self.lib.kmers_host_free(self.handle, C.c_void_p(ptr))

    def h2d(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.kmers_memcpy_h2d(self.handle, C.c_void_p(dptr), arr.ctypes.data_as(C.c_void_p),
                                             arr.nbytes), "kmers_memcpy_h2d")

    def d2h(self, arr, dptr):
        self.check(self.lib.kmers_memcpy_d2h(self.handle, arr.ctypes.data_as(C.c_void_p), C.c_void_p(dptr),
                                             arr.nbytes), "kmers_memcpy_d2h")

    def sync(self):
        res = _capi.Result()
        rc = self.lib.kmers_sync(self.handle, C.byref(res))
        return rc, res


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


# --------------------------------------------------------------------------------------------
# sequences
class LongSequence:
    """BioSequences.LongSequence{A}: `.data` = little-endian packed UInt64 words, `.len` symbols."""

    def __init__(self, alphabet, source="", length=None):
        self.alphabet = alphabet
        if isinstance(source, (str, bytes)):
            text = source.decode() if isinstance(source, bytes) else source
            enc = _ENC2 if alphabet.bits == 2 else _ENC4
            try:
                codes = np.fromiter((enc[c] for c in text.upper()), dtype=np.uint64, count=len(text))
            except KeyError as e:
                raise EncodeError(alphabet, e.args[0], None, None)
            self.len = len(text)
            per = 64 // alphabet.bits
            nw = (self.len + per - 1) // per
            pad = np.zeros(nw * per, dtype=np.uint64)
            pad[:self.len] = codes
            shifts = (np.arange(per, dtype=np.uint64) * np.uint64(alphabet.bits))
            self.data = np.bitwise_or.reduce(pad.reshape(nw, per) << shifts, axis=1).astype(np.uint64) \
                if nw else np.zeros(0, dtype=np.uint64)
        else:
            self.data = np.ascontiguousarray(source, dtype=np.uint64)
            self.len = int(length)
        self._dev = None  # (ctx, device pointer)

    def __len__(self):
        return self.len

    @property
    def src_bits(self):
        return self.alphabet.bits

    def __str__(self):
        inv = _decode_table(self.alphabet)
        b = self.alphabet.bits
        return "".join(inv[(int(self.data[(i * b) >> 6]) >> ((i * b) & 63)) & ((1 << b) - 1)]
                       for i in range(self.len))

    def __repr__(self):
        s = str(self) if self.len <= 60 else str(LongSequence(self.alphabet, self.data, 57)) + "..."
        return f"{self.len}nt {self.alphabet.kind} Sequence:\n{s}"

    def device_words(self, ctx):
        """HBM-resident copy of `.data` (uploaded once, reused by every iterator pass)."""
        if self._dev is None or self._dev[0] is not ctx:
            ptr = ctx.alloc(max(8, self.data.nbytes + 8))
            if self.data.nbytes:
                ctx.h2d(ptr, self.data)
            self._dev = (ctx, ptr)
        return self._dev[1]

    def __del__(self):
        try:
            if self._dev is not None:
                self._dev[0].free(self._dev[1])
        except Exception:
            pass


class _LongFamily:
    def __init__(self, fam):
        self.fam = fam

    def __getitem__(self, bits):
        alph = self.fam[bits]
        return lambda source="", length=None: LongSequence(alph, source, length)


LongDNA = _LongFamily(DNAAlphabet)
LongRNA = _LongFamily(RNAAlphabet)


class AsciiSource:
    """A String / Vector{UInt8} source (AsciiEncode, src/construction.jl:94-95): one byte per symbol."""
    alphabet = None
    src_bits = 8

    def __init__(self, source):
        raw = source.encode("latin-1") if isinstance(source, str) else bytes(source)
        self.len = len(raw)
        pad = (-self.len) % 8 + 8  # whole 8-byte words plus one spare word
        self.data = np.frombuffer(raw + b"\0" * pad, dtype=np.uint8)
        self._dev = None

    def __len__(self):
        return self.len

    def __str__(self):
        return self.data[:self.len].tobytes().decode("latin-1")

    device_words = LongSequence.device_words
    __del__ = LongSequence.__del__


class SymbolVector:
    """A `Vector{DNA}` / `Vector{RNA}` -- any collection of nucleotide symbols that is neither a BioSequence nor text:
    the reference iterates it through GenericRecoding (src/construction.jl:90-98, FwKmers.jl:80-86), symbol by symbol
    through `BioSequences.encode` of the kmer alphabet.  One byte per symbol holding its BioSymbols value."""
    src_bits = 8
    symbols = True

    def __init__(self, kind, source):
        self.alphabet = _Alphabet(kind, 4)
        if isinstance(source, str):
            try:
                raw = bytes(_ENC4[c] for c in source.upper().replace("U", "T"))
            except KeyError as e:
                raise EncodeError(self.alphabet, e.args[0], None, None)
        else:
            raw = bytes(int(v) for v in source)
        self.len = len(raw)
        self.data = np.frombuffer(raw + b"\0" * ((-self.len) % 8 + 8), dtype=np.uint8)
        self._dev = None

    def __len__(self):
        return self.len

    def __str__(self):
        inv = _decode_table(self.alphabet)
        return "".join(inv[int(v)] for v in self.data[:self.len])

    device_words = LongSequence.device_words
    __del__ = LongSequence.__del__


def _as_sequence(s):
    if isinstance(s, (LongSequence, AsciiSource, SymbolVector)):
        return s
    if isinstance(s, (str, bytes, bytearray, np.ndarray)):
        return AsciiSource(s)
    raise UnsupportedError(f"unsupported source type {type(s).__name__}")


# --------------------------------------------------------------------------------------------
# kmers
def n_coding_elements(K, bits):
    return (K * bits + 63) // 64


class Kmer:
    """Kmer{A,K,N} value (src/kmer.jl:32-44): `.data` = N UInt64, first symbol most significant."""
    __slots__ = ("alphabet", "K", "data")

    def __init__(self, alphabet, K_or_text, data=None):
        self.alphabet = alphabet
        if data is None:
            text = K_or_text
            enc = _ENC2 if alphabet.bits == 2 else _ENC4
            v = 0
            for ch in text.upper():
                if ch not in enc:
                    raise EncodeError(alphabet, ch, None, None)
                v = (v << alphabet.bits) | enc[ch]
            self.K = len(text)
            N = n_coding_elements(self.K, alphabet.bits)
            self.data = tuple((v >> (64 * (N - 1 - i))) & MASK64 for i in range(N))
        else:
            self.K = int(K_or_text)
            self.data = tuple(int(x) for x in data)

    def __len__(self):
        return self.K

    def __str__(self):
        inv = _decode_table(self.alphabet)
        b = self.alphabet.bits
        v = 0
        for w in self.data:
            v = (v << 64) | w
        return "".join(inv[(v >> (b * (self.K - 1 - t))) & ((1 << b) - 1)] for t in range(self.K))

    def __repr__(self):
        return f"{self.alphabet.kind} {self.K}-mer:\n{self}"

    def _key(self):
        return (self.alphabet.bits, self.K, self.data)

    def __eq__(self, other):
        return isinstance(other, Kmer) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())

    def __lt__(self, other):  # same-K tuple comparison, src/kmer.jl:176-178,200
        if self.alphabet.bits != other.alphabet.bits or self.K != other.K:
            raise KmersError("kmers of different type are not ordered here")
        return self.data < other.data

    def __le__(self, other):
        return self == other or self < other


class _KmerFamily:
    def __init__(self, alphabet_family):
        self.fam = alphabet_family

    def __getitem__(self, K):
        alph = self.fam[2]

        def make(text):
            k = Kmer(alph, text)
            if k.K != K:
                raise KmersError("Length of sequence must be K elements to build Kmer")
            return k
        return make


DNAKmer = _KmerFamily(DNAAlphabet)
RNAKmer = _KmerFamily(RNAAlphabet)


def mer(text, flag="d"):
    """`mer"TAG"d` / `mer"UAG"r` literal (src/construction.jl:360-374); 2-bit alphabets."""
    return Kmer(DNAAlphabet[2] if flag == "d" else RNAAlphabet[2], text)


class KmerArray:
    """Vector{Kmer{A,K,N}}: an (n, N) uint64 array in exactly the reference's memory layout."""

    def __init__(self, alphabet, K, words):
        self.alphabet, self.K = alphabet, K
        self.N = n_coding_elements(K, alphabet.bits)
        self.words = np.ascontiguousarray(words, dtype=np.uint64).reshape(-1, max(1, self.N))

    def __len__(self):
        return self.words.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return KmerArray(self.alphabet, self.K, self.words[i])
        return Kmer(self.alphabet, self.K, self.words[i])

    def __iter__(self):
        for row in self.words:
            yield Kmer(self.alphabet, self.K, row)

    def __eq__(self, other):
        if isinstance(other, KmerArray):
            return (self.alphabet.bits, self.K) == (other.alphabet.bits, other.K) and \
                np.array_equal(self.words, other.words)
        return list(self) == list(other)

    def tolist(self):
        return list(self)


# --------------------------------------------------------------------------------------------
# iterators
def _raise_encode(alphabet, seq, res):
    if getattr(seq, "symbols", False):  # EncodeError(A, symbol) from BioSequences.encode (kmer.jl:445-448)
        sym = _decode_table(seq.alphabet)[res.err_enc] if res.err_enc < 16 else f"0x{res.err_enc:02x}"
    elif seq.src_bits == 8:  # EncodeError(A, repr(byte)), FwKmers.jl:124-126
        sym = f"0x{res.err_enc:02x} (Char {chr(res.err_enc)!r})"
    else:
        sym = _decode_table(seq.alphabet)[res.err_enc] if seq.src_bits == 4 else "?"
    raise EncodeError(alphabet, sym, int(res.err_pos), int(res.err_enc))


class _Bound:
    """A parametrised iterator type, e.g. `CanonicalKmers[DNAAlphabet[2], 31]`: call it with a sequence."""

    def __init__(self, cls, params):
        self.cls, self.params = cls, params

    def __call__(self, seq, ctx=None, **kw):
        return self.cls(*self.cls._expand(self.params), seq, ctx=ctx, **kw)


class _Parametric(type):
    """`Iterator[params](seq)` stands for Julia's `Iterator{params}(seq)`."""

    def __getitem__(cls, params):
        if not isinstance(params, tuple):
            params = (params,)
        return _Bound(cls, params)


class AbstractKmerIterator(metaclass=_Parametric):
    CHUNK = 1 << 20  # kmers per pass of the chunk-buffered `for x in it`

    def __init__(self, alphabet, K, seq, ctx=None, stride=1):
        if not isinstance(K, int):
            raise KmersError("K must be an Int")          # FwKmers.jl:32
        if K < 1:
            raise KmersError("K must be at least 1")      # FwKmers.jl:33
        if stride < 1:
            raise KmersError("J must be at least 1")      # SpacedKmers.jl:30
        self.alphabet, self.K, self.J = alphabet, K, stride
        self.seq = _as_sequence(seq)
        self.ctx = ctx or default_context()
        self.N = n_coding_elements(K, alphabet.bits)
        if not self.ctx.lib.kmers_supported(self.seq.src_bits, alphabet.bits, K, stride):
            raise UnsupportedError(
                f"Kmer{{{alphabet},{K}}} from {self.seq.alphabet or 'ASCII bytes'} is outside the kernels' coverage")

    # Base.eltype (src/iterators/common.jl:13-15)
    @property
    def eltype(self):
        return ("Kmer", self.alphabet, self.K, self.N)

    def __len__(self):  # FwKmers.jl:40-43, SpacedKmers.jl:38-42
        return int(self.ctx.lib.kmers_count(self.seq.len, self.K, self.J))

    def _view(self, first_base, n_bases):
        return _capi.Seq(self.seq.device_words(self.ctx), n_bases, first_base, first_base,
                         self.seq.src_bits, 2 if getattr(self.seq, "symbols", False) else (1 if self.alphabet.kind == "RNA" else 0))

    def _wrap(self, arrays, lo, hi):
        raise NotImplementedError

    def _run(self, view, n):
        raise NotImplementedError

    def _range(self, start_kmer, n):
        """Compute kmers [start_kmer, start_kmer+n) of the iteration -> host arrays."""
        first = start_kmer * self.J
        nb = (n - 1) * self.J + self.K
        return self._run(self._view(first, nb), n)

    def collect(self):
        n = len(self)
        out, res = self._range(0, n) if n else (self._empty(), None)
        if res is not None and res.status == _capi.E_ENCODE:
            _raise_encode(self.alphabet, self.seq, res)
        return self._wrap(out)

    def __iter__(self):
        """Chunk-buffered iteration with the reference's throw point: elements whose windows end before the first ambiguous
        symbol are yielded, then EncodeError is raised.  Chunk c + 1 is launched (KMERS_ASYNC) and its copy into page-locked
        memory enqueued BEFORE the loop is handed chunk c; kmers_sync is called when chunk c is used up (_ChunkPipe: the pipeline
        julia/KmersHIP.jl's GPUIterator runs)."""
        total = self._units()
        if total == 0:
            return
        pipe = _ChunkPipe(self, min(self.CHUNK, total))
        try:
            u0, m = 0, min(self.CHUNK, total)
            pipe.enqueue(0, u0, m)
            slot = 0
            while m:
                rc, res = self.ctx.sync()                      # the chunk in flight, and only it
                if rc == _capi.E_ENCODE:
                    good = self._yielded_before(int(res.err_pos) - 1) - u0
                    if good > 0:
                        out, _ = self._range(u0, good)
                        yield from self._elements(self._wrap(out))
                    _raise_encode(self.alphabet, self.seq, res)
                self.ctx.check(rc, type(self).__name__)
                out = pipe.take(slot, int(res.n_out) if self._counted_by_sync and self.seq.src_bits != 2 else m)   # (a 2-bit source drops nothing)
                u1 = u0 + m
                m = min(self.CHUNK, total - u1)
                u0, slot = u1, 1 - slot
                if m:
                    pipe.enqueue(slot, u0, m)                  # on its way while the caller loops over `out`
                yield from self._elements(self._wrap(out))
        finally:
            pipe.close()

    _counted_by_sync = False   # UnambiguousKmers: the number of elements of a chunk is known when it has run

    def _units(self):
        """what a chunk is counted in: elements (UnambiguousKmers: candidate windows)"""
        return len(self)

    def _widths(self):
        """words per element of each output array of a chunk"""
        return [self.N]

    def _enqueue(self, view, n, ptrs, res):
        raise NotImplementedError

    def _yielded_before(self, bad0):
        """Number of elements iterate() yields before it touches symbol bad0 (0-based)."""
        if self.J >= self.K:  # element m touches only [mJ, mJ+K): the bad symbol sits in element bad0 // J
            return bad0 // self.J
        if bad0 < self.K:
            return 0
        return (bad0 - self.K) // self.J + 1  # rolling: element m needs every symbol below mJ+K

    def _elements(self, wrapped):
        return iter(wrapped)

    def _host_out(self, n, width):
        return np.zeros((n, width), dtype=np.uint64)

    def _dev_out(self, n, width):
        ptr = self.ctx.alloc(max(8, n * width * 8))
        return ptr


class _ChunkPipe:
    """The buffers of a chunk-buffered iteration: per output array two chunks in HBM and two in page-locked host memory, taking
    turns.  enqueue() launches units [u0, u0 + n) into one slot and enqueues the copies of its arrays; nothing waits."""

    def __init__(self, it, cap):
        self.it, self.ctx, self.cap = it, it.ctx, cap
        self.widths = it._widths()
        self.dev = [[self.ctx.alloc(max(8, cap * w * 8), lone_output=len(self.widths) == 1) for w in self.widths] for _ in range(2)]
        self.host = [[self.ctx.host_alloc(max(8, cap * w * 8)) for w in self.widths] for _ in range(2)]
        self.res = _capi.Result()

    def enqueue(self, slot, u0, n):
        it = self.it
        view = it._view(u0 * it.J, (n - 1) * it.J + it.K)
        rc = it._enqueue(view, n, self.dev[slot], self.res)
        self.ctx.check(rc, type(it).__name__)
        for d, h, w in zip(self.dev[slot], self.host[slot], self.widths):
            self.ctx.check(self.ctx.lib.kmers_memcpy_d2h_async(self.ctx.handle, C.c_void_p(h), C.c_void_p(d), n * w * 8), "kmers_memcpy_d2h_async")

    def take(self, slot, n):
        """the first n elements of the slot's arrays, copied out of the page-locked buffers (which the chunk after next overwrites)"""
        outs = []
        for h, w in zip(self.host[slot], self.widths):
            a = np.ctypeslib.as_array(C.cast(C.c_void_p(h), C.POINTER(C.c_uint64)), shape=(max(1, self.cap * w),))
            outs.append(a[:n * w].reshape(n, w).copy())
        return outs

    def close(self):
        self.ctx.sync()
        for slot in self.dev:
            for p in slot:
                self.ctx.free(p)
        for slot in self.host:
            for p in slot:
                self.ctx.host_free(p)
        self.dev, self.host = [], []


def _call_with_device_outputs(it, n, widths, call):
    """Allocate device outputs, run `call(ptrs)`, copy back; returns (host arrays, Result)."""
    ctx = it.ctx
    # the ONLY output array of a launch is allocated by that role (kmers_dev_alloc_role: its second half in another region class
    # of HBM than its first, and the launch writes it through two windows)
    ptrs = [ctx.alloc(max(8, n * w * 8), lone_output=len(widths) == 1) for w in widths]
    res = _capi.Result()
    try:
        rc = call(ptrs, res)
        ctx.check(rc, type(it).__name__)
        outs = []
        for p, w in zip(ptrs, widths):
            a = np.zeros((n, w), dtype=np.uint64)
            if rc == _capi.OK and n:
                ctx.d2h(a, p)
            outs.append(a)
    finally:
        for p in ptrs:
            ctx.free(p)
    return outs, res


class FwKmers(AbstractKmerIterator):
    """FwKmers{A,K}(seq) (src/iterators/FwKmers.jl:28-115)."""
    @staticmethod
    def _expand(p):
        return p  # (A, K)

    def _empty(self):
        return [np.zeros((0, self.N), dtype=np.uint64)]

    def _run(self, view, n):
        lib, h = self.ctx.lib, self.ctx.handle
        return _call_with_device_outputs(
            self, n, [self.N],
            lambda ptrs, res: lib.kmers_fw(h, C.byref(view), self.K, self.alphabet.bits, ptrs[0], None,
                                           _capi.MEM_DEVICE, C.byref(res)))

    def _wrap(self, out):
        return KmerArray(self.alphabet, self.K, out[0])

    def _enqueue(self, view, n, ptrs, res):
        return self.ctx.lib.kmers_fw(self.ctx.handle, C.byref(view), self.K, self.alphabet.bits, ptrs[0], None,
                                     _capi.MEM_DEVICE | _capi.ASYNC, C.byref(res))


class FwRvIterator(AbstractKmerIterator):
    """FwRvIterator{A,K}(seq): (forward, reverse_complement) pairs (CanonicalKmers.jl:25-144)."""
    @staticmethod
    def _expand(p):
        return p

    def _empty(self):
        return [np.zeros((0, self.N), dtype=np.uint64)] * 2

    def _run(self, view, n):
        lib, h = self.ctx.lib, self.ctx.handle
        return _call_with_device_outputs(
            self, n, [self.N, self.N],
            lambda ptrs, res: lib.kmers_fw(h, C.byref(view), self.K, self.alphabet.bits, ptrs[0], ptrs[1],
                                           _capi.MEM_DEVICE, C.byref(res)))

    def _wrap(self, out):
        return (KmerArray(self.alphabet, self.K, out[0]), KmerArray(self.alphabet, self.K, out[1]))

    def _elements(self, wrapped):
        return zip(iter(wrapped[0]), iter(wrapped[1]))

    def _widths(self):
        return [self.N, self.N]

    def _enqueue(self, view, n, ptrs, res):
        return self.ctx.lib.kmers_fw(self.ctx.handle, C.byref(view), self.K, self.alphabet.bits, ptrs[0], ptrs[1],
                                     _capi.MEM_DEVICE | _capi.ASYNC, C.byref(res))

    def collect(self):
        fw, rv = super().collect()
        return list(zip(fw, rv))


class CanonicalKmers(AbstractKmerIterator):
    """CanonicalKmers{A,K}(seq): fw < rv ? fw : rv (CanonicalKmers.jl:199-225)."""
    @staticmethod
    def _expand(p):
        return p

    def _empty(self):
        return [np.zeros((0, self.N), dtype=np.uint64)]

    def _run(self, view, n):
        lib, h = self.ctx.lib, self.ctx.handle
        return _call_with_device_outputs(
            self, n, [self.N],
            lambda ptrs, res: lib.kmers_canonical(h, C.byref(view), self.K, self.alphabet.bits, ptrs[0], None, 0,
                                                  _capi.MEM_DEVICE, C.byref(res)))

    def _wrap(self, out):
        return KmerArray(self.alphabet, self.K, out[0])

    def _enqueue(self, view, n, ptrs, res):
        return self.ctx.lib.kmers_canonical(self.ctx.handle, C.byref(view), self.K, self.alphabet.bits, ptrs[0], None, 0,
                                            _capi.MEM_DEVICE | _capi.ASYNC, C.byref(res))

    def collect_with_hashes(self, seed=0):
        """collect(it) together with fx_hash.(kmers, seed) from the same pass (one fused kernel)."""
        n = len(self)
        if n == 0:
            return KmerArray(self.alphabet, self.K, np.zeros((0, self.N), np.uint64)), np.zeros(0, np.uint64)
        lib, h = self.ctx.lib, self.ctx.handle
        view = self._view(0, self.seq.len)
        out, res = _call_with_device_outputs(
            self, n, [self.N, 1],
            lambda ptrs, res: lib.kmers_canonical(h, C.byref(view), self.K, self.alphabet.bits, ptrs[0], ptrs[1],
                                                  seed, _capi.MEM_DEVICE, C.byref(res)))
        if res.status == _capi.E_ENCODE:
            _raise_encode(self.alphabet, self.seq, res)
        return KmerArray(self.alphabet, self.K, out[0]), out[1].reshape(-1)


class SpacedKmers(AbstractKmerIterator):
    """SpacedKmers{A,K,J}(seq) (src/iterators/SpacedKmers.jl:23-139), strict semantics."""

    def __init__(self, alphabet, K, J, seq, ctx=None):
        if not isinstance(J, int):
            raise KmersError("J must be an Int")  # SpacedKmers.jl:29
        super().__init__(alphabet, K, seq, ctx=ctx, stride=J)

    @staticmethod
    def _expand(p):
        return p  # (A, K, J)

    def _empty(self):
        return [np.zeros((0, self.N), dtype=np.uint64)]

    def _run(self, view, n):
        lib, h = self.ctx.lib, self.ctx.handle
        return _call_with_device_outputs(
            self, n, [self.N],
            lambda ptrs, res: lib.kmers_spaced(h, C.byref(view), self.K, self.J, self.alphabet.bits, ptrs[0],
                                               _capi.MEM_DEVICE, C.byref(res)))

    def _wrap(self, out):
        return KmerArray(self.alphabet, self.K, out[0])

    def _enqueue(self, view, n, ptrs, res):
        return self.ctx.lib.kmers_spaced(self.ctx.handle, C.byref(view), self.K, self.J, self.alphabet.bits, ptrs[0],
                                         _capi.MEM_DEVICE | _capi.ASYNC, C.byref(res))


class UnambiguousKmers(AbstractKmerIterator):
    """UnambiguousKmers{A,K}(seq): (kmer, start) of every window without ambiguous symbols
    (src/iterators/UnambiguousKmers.jl:29-148).  SizeUnknown unless the source is 2-bit (:33-37).
    `stride` > 1 keeps the windows on the stride lattice (BASELINE.json config 5 skip variant)."""

    def __init__(self, alphabet, K, seq, ctx=None, stride=1):
        if alphabet.bits != 2:
            raise KmersError("UnambiguousKmers needs a 2-bit kmer alphabet")  # A <: TwoBit, :29
        super().__init__(alphabet, K, seq, ctx=ctx, stride=1)
        self.lattice = stride

    @staticmethod
    def _expand(p):
        return p

    def __len__(self):
        if self.seq.src_bits == 2 and self.lattice == 1:
            return super().__len__()
        raise TypeError("UnambiguousKmers over a 4-bit source is SizeUnknown")  # :33

    def collect(self):
        lib, ctx = self.ctx.lib, self.ctx
        view = self._view(0, self.seq.len)
        res = _capi.Result()
        rc = ctx.check(lib.kmers_unambiguous(ctx.handle, C.byref(view), self.K, self.lattice, None, None, 0,
                                             _capi.MEM_DEVICE, C.byref(res)), "kmers_unambiguous(count)")
        if rc == _capi.E_ENCODE:
            _raise_encode(self.alphabet, self.seq, res)
        n = int(res.n_out)
        km = np.zeros((n, self.N), dtype=np.uint64)
        st = np.zeros(n, dtype=np.int64)
        if n:
            pk, ps = ctx.alloc(km.nbytes), ctx.alloc(st.nbytes)
            try:
                ctx.check(lib.kmers_unambiguous(ctx.handle, C.byref(view), self.K, self.lattice, pk, ps, n,
                                                _capi.MEM_DEVICE, C.byref(res)), "kmers_unambiguous")
                ctx.d2h(km, pk)
                ctx.d2h(st, ps)
            finally:
                ctx.free(pk)
                ctx.free(ps)
        return list(zip(KmerArray(self.alphabet, self.K, km), st.tolist()))

    # chunk-buffered `for (kmer, start) in it` (UnambiguousKmers.jl:59-62): chunks of candidate windows, the number kept known at
    # the chunk's kmers_sync; starts are those of the whole sequence (the chunk view's index_origin)
    _counted_by_sync = True

    def _units(self):
        return max(0, self.seq.len - self.K + 1)

    def _widths(self):
        return [self.N, 1]

    def _enqueue(self, view, n, ptrs, res):
        return self.ctx.lib.kmers_unambiguous(self.ctx.handle, C.byref(view), self.K, 1, ptrs[0], ptrs[1], n,
                                              _capi.MEM_DEVICE | _capi.ASYNC, C.byref(res))

    def _range(self, start, n):
        """the kept windows among candidate starts [start, start + n), computed now (the clean prefix before an EncodeError)"""
        lib, ctx = self.ctx.lib, self.ctx
        view = self._view(start, n + self.K - 1)
        out, res = _call_with_device_outputs(
            self, n, [self.N, 1],
            lambda ptrs, r: lib.kmers_unambiguous(ctx.handle, C.byref(view), self.K, 1, ptrs[0], ptrs[1], n, _capi.MEM_DEVICE, C.byref(r)))
        kept = int(res.n_out) if res.status == _capi.OK else 0
        return [out[0][:kept], out[1][:kept]], res

    def _wrap(self, out):
        return (KmerArray(self.alphabet, self.K, out[0]), out[1].reshape(-1).astype(np.int64))

    def _elements(self, wrapped):
        return zip(iter(wrapped[0]), wrapped[1].tolist())

    def __iter__(self):
        if self.lattice != 1:   # (the lattice variant of BASELINE config 5 is not an iterator of the reference)
            return iter(self.collect())
        return AbstractKmerIterator.__iter__(self)


def _alias(cls, fam):
    class _A:
        def __getitem__(self, params):
            if not isinstance(params, tuple):
                params = (params,)
            return _Bound(cls, (fam[2],) + params)
    return _A()


FwDNAMers = _alias(FwKmers, DNAAlphabet)                   # FwKmers.jl:48-49
FwRNAMers = _alias(FwKmers, RNAAlphabet)                   # FwKmers.jl:51-52
FwRvDNAIterator = _alias(FwRvIterator, DNAAlphabet)        # CanonicalKmers.jl:35-36
FwRvRNAIterator = _alias(FwRvIterator, RNAAlphabet)        # CanonicalKmers.jl:38-39
CanonicalDNAMers = _alias(CanonicalKmers, DNAAlphabet)     # CanonicalKmers.jl:214-215
CanonicalRNAMers = _alias(CanonicalKmers, RNAAlphabet)     # CanonicalKmers.jl:217-218
UnambiguousDNAMers = _alias(UnambiguousKmers, DNAAlphabet)  # UnambiguousKmers.jl:53-54
UnambiguousRNAMers = _alias(UnambiguousKmers, RNAAlphabet)  # UnambiguousKmers.jl:56-57
SpacedDNAMers = _alias(SpacedKmers, DNAAlphabet)           # SpacedKmers.jl:46-47
SpacedRNAMers = _alias(SpacedKmers, RNAAlphabet)           # SpacedKmers.jl:49-50


def each_codon(*args, ctx=None):
    """each_codon(s::BioSequence) / each_codon(DNA | RNA, s) (SpacedKmers.jl:77-81)."""
    if len(args) == 2:
        kind, seq = args
        fam = RNAAlphabet if kind in ("RNA", RNA) else DNAAlphabet
    else:
        (seq,) = args
        fam = RNAAlphabet if seq.alphabet.kind == "RNA" else DNAAlphabet
    return SpacedKmers(fam[2], 3, 3, seq, ctx=ctx)


DNA, RNA = "DNA", "RNA"


# The dispatch policy of julia/KmersHIP.jl (its header comment): Base.collect of an iterator goes to the device only from
# MIN_BASES symbols on (a call costs ~16-25 us plus two PCIe hops, the reference iterates at ~1 ns per symbol: the docstring
# case collect(FwDNAMers{3}("AGCGTATA")), src/iterators/FwKmers.jl:14-22, must stay a CPU call there).  This mirror has no
# reference implementation to hand a short sequence to -- and no CPU fallback of its own, by construction -- so ITS collect
# always runs on the device; gpu_dispatch() reports what the Julia binding would decide, and is what the tests pin.
MIN_BASES = int(os.environ.get("KMERS_HIP_MIN_BASES", "100000"))


def gpu_dispatch(it):
    """True iff `Base.collect(it)` of the Julia binding runs on the device (KmersHIP.gpu_dispatch)."""
    return len(it.seq) >= MIN_BASES


def gpu_collect(it):
    """KmersHIP.gpu_collect: the bulk form on the device, whatever the length."""
    return it.collect()


def collect(it):
    return it.collect()


def _build_pool(records):
    """Records -> (sequences, pool array, spans, pool symbols, source bits): LongSequence data words are
    concatenated (every record starts on a word boundary), byte records are joined."""
    recs = [_as_sequence(r) for r in records]
    src_bits = recs[0].src_bits if recs else 2
    if any(r.src_bits != src_bits for r in recs):
        raise UnsupportedError("all records of a batch must share one source type")
    spans = (_capi.Span * max(len(recs), 1))()
    if src_bits == 8:
        pool = np.concatenate([r.data[:r.len] for r in recs] + [np.zeros(16, np.uint8)]) if recs else np.zeros(16, np.uint8)
        pos = 0
        for i, r in enumerate(recs):
            spans[i] = _capi.Span(pos, r.len)
            pos += r.len
        n_pool = pos
        pool = np.ascontiguousarray(pool)
    else:
        per = 64 // src_bits
        pool = np.concatenate([r.data for r in recs] + [np.zeros(1, np.uint64)]) if recs else np.zeros(1, np.uint64)
        w = 0
        for i, r in enumerate(recs):
            spans[i] = _capi.Span(w * per, r.len)
            w += len(r.data)
        n_pool = w * per
    return recs, pool, spans, n_pool, src_bits


def collect_batch(iterator, records, hashes=False, seed=0, ctx=None, skip_ambiguous=False):
    """Elements of `iterator(record)` for every record of a batch, concatenated in record order, from
    ONE launch (include/kmers_hip.h `kmers_batch`): the reference's `for record in reader ...
    CanonicalDNAMers{K}(sequence(record))` loop (docs/src/minhash.md:31-35) without the per-call cost.

    iterator: a parametrised iterator type, e.g. `CanonicalDNAMers[31]`, `FwDNAMers[21]`,
              `FwRvIterator[DNAAlphabet[2], 31]`, `SpacedDNAMers[3, 3]` (each_codon of every record:
              `kmers_batch_spaced`).
    records:  LongSequences of one alphabet, or str / bytes records (ASCII).
    Returns (first, second, offsets): FwKmers / SpacedKmers -> (kmers, None, offsets); FwRvIterator -> (kmers,
    reverse complements, offsets); CanonicalKmers -> (kmers, fx_hash values if `hashes` else None,
    offsets).  Record i owns elements offsets[i]:offsets[i+1].  skip_ambiguous=True (KMERS_BATCH_SKIP): windows over
    symbols the kmer alphabet cannot encode are written as all-ones instead of raising EncodeError."""
    ctx = ctx or default_context()
    cls, params = getattr(iterator, "cls", None), getattr(iterator, "params", None)
    if cls not in (FwKmers, FwRvIterator, CanonicalKmers, SpacedKmers):
        raise UnsupportedError("collect_batch(FwKmers / FwRvIterator / CanonicalKmers [alphabet, K] or SpacedKmers [alphabet, K, J], records)")
    if cls is SpacedKmers:
        alphabet, K, J = cls._expand(params)
        if not isinstance(J, int) or J < 1:
            raise KmersError("J must be at least 1")      # SpacedKmers.jl:29-30
    else:
        alphabet, K = cls._expand(params)
    recs, pool, spans, n_pool, src_bits = _build_pool(records)
    seq = _capi.Seq(pool.ctypes.data, n_pool, 0, 0, src_bits, 1 if alphabet.kind == "RNA" else 0)
    res = _capi.Result()
    offsets = np.zeros(len(recs) + 1, dtype=np.uint64)
    mode = _capi.BATCH_CANONICAL if cls is CanonicalKmers else _capi.BATCH_FW
    mem = _capi.MEM_HOST | (_capi.BATCH_SKIP if skip_ambiguous else 0)
    N = n_coding_elements(K, alphabet.bits)
    poff = offsets.ctypes.data_as(C.c_void_p)

    def call(pa, pb, cap):
        if cls is SpacedKmers:
            return ctx.lib.kmers_batch_spaced(ctx.handle, C.byref(seq), spans, len(recs), K, J, alphabet.bits, pa, poff, cap, mem, C.byref(res))
        return ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans, len(recs), mode, K, alphabet.bits, pa, pb, seed & MASK64, poff, cap, mem,
                                   C.byref(res))
    ctx.check(call(None, None, 0), "kmers_batch")
    total = int(res.n_out)
    first = np.zeros((max(total, 1), N), dtype=np.uint64)
    want_second = cls is FwRvIterator or (cls is CanonicalKmers and hashes)
    second = np.zeros((max(total, 1), 1 if cls is CanonicalKmers else N), dtype=np.uint64) if want_second else None
    rc = call(first.ctypes.data_as(C.c_void_p), second.ctypes.data_as(C.c_void_p) if want_second else None, total)
    if rc == _capi.E_ENCODE:
        bad = recs[int(res.n_out)]
        _raise_encode(alphabet, bad, res)
    ctx.check(rc, "kmers_batch")
    kmers = KmerArray(alphabet, K, first[:total])
    if cls is FwRvIterator:
        return kmers, KmerArray(alphabet, K, second[:total]), offsets.astype(np.int64)
    return kmers, (second[:total, 0] if want_second else None), offsets.astype(np.int64)


def reducer(it):
    """The reducer of the reference's throughput benchmark (test/benchmark.jl:9-15):
    `y = 0; for i in it; y ⊻= extract_kmer(i).data[1]; end` -- fused on the device, nothing materialised.
    Works for FwKmers, FwRvIterator (first of the pair), CanonicalKmers, SpacedKmers and UnambiguousKmers."""
    if isinstance(it, UnambiguousKmers):
        code, stride = _capi.ITER_UNAMBIGUOUS, it.lattice
    elif isinstance(it, SpacedKmers):
        code, stride = _capi.ITER_SPACED, it.J
    elif isinstance(it, CanonicalKmers):
        code, stride = _capi.ITER_CANONICAL, 1
    elif isinstance(it, (FwKmers, FwRvIterator)):
        code, stride = _capi.ITER_FW, 1
    else:
        raise UnsupportedError("reducer(FwKmers | FwRvIterator | CanonicalKmers | SpacedKmers | UnambiguousKmers)")
    val = C.c_uint64()
    res = _capi.Result()
    view = it._view(0, it.seq.len)
    rc = it.ctx.lib.kmers_reduce_xor_iter(it.ctx.handle, C.byref(view), it.K, it.alphabet.bits, code, stride, C.byref(val),
                                          _capi.MEM_DEVICE, C.byref(res))
    if rc == _capi.E_ENCODE:
        _raise_encode(it.alphabet, it.seq, res)
    it.ctx.check(rc, "kmers_reduce_xor_iter")
    return val.value


def sketch_batch(f, iterator, records, s, seed=0, ctx=None):
    """[MinHash.sketch(fx_hash, CanonicalKmers{A,K}(r), s) for r in records] from one launch sequence
    (`kmers_minhash_batch`): a list of ascending uint64 arrays, one per record (docs/src/minhash.md:31-41)."""
    ctx = ctx or default_context()
    cls, params = getattr(iterator, "cls", None), getattr(iterator, "params", None)
    if f is not fx_hash or cls is not CanonicalKmers:
        raise UnsupportedError("sketch_batch(fx_hash, CanonicalKmers[alphabet, K], records, s)")
    alphabet, K = cls._expand(params)
    recs, pool, spans, n_pool, src_bits = _build_pool(records)
    if not recs:
        return []
    seq = _capi.Seq(pool.ctypes.data, n_pool, 0, 0, src_bits, 1 if alphabet.kind == "RNA" else 0)
    out = np.zeros((len(recs), int(s)), dtype=np.uint64)
    counts = np.zeros(len(recs), dtype=np.uint64)
    res = _capi.Result()
    rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(seq), spans, len(recs), K, alphabet.bits, seed & MASK64, int(s),
                                     out.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p), _capi.MEM_HOST, C.byref(res))
    if rc == _capi.E_ENCODE:
        _raise_encode(alphabet, recs[int(res.n_out)], res)
    ctx.check(rc, "kmers_minhash_batch")
    return [out[i, :int(counts[i])].copy() for i in range(len(recs))]


def sketch(f, it, s, seed=0):
    """MinHash.sketch(fx_hash, CanonicalKmers{A,K}(seq), s) (docs/src/minhash.md:31-35): the s
    smallest distinct fx_hash values of the canonical kmers, ascending, from one fused pass."""
    if f is not fx_hash or not isinstance(it, CanonicalKmers):
        raise UnsupportedError("the fused sketch is sketch(fx_hash, CanonicalKmers, s)")
    out = np.zeros(max(int(s), 1), dtype=np.uint64)
    res = _capi.Result()
    view = it._view(0, it.seq.len)
    rc = it.ctx.check(it.ctx.lib.kmers_minhash(it.ctx.handle, C.byref(view), it.K, it.alphabet.bits, seed & MASK64,
                                                int(s), out.ctypes.data_as(C.c_void_p), _capi.MEM_DEVICE,
                                                C.byref(res)), "kmers_minhash")
    if rc == _capi.E_ENCODE:
        _raise_encode(it.alphabet, it.seq, res)
    return out[:int(res.n_out)]


def minimizers(it, W, stride=1, mode=0):
    """Minimizers over the kmers of a FwKmers iterator (docs/src/replacements.md:33-51,
    test/benchmark.jl:96-119): element j = the kmer with the smallest fx_hash among the W kmers
    starting at symbol 1 + j*stride.  mode 0 = the reference's published example literally
    (symbols are shifted into the running minimum), mode 1 = true sliding-window minimum."""
    if not isinstance(it, FwKmers):
        raise UnsupportedError("minimizers(FwKmers, W)")
    span = it.K + W - 1
    n = 0 if it.seq.len < span else (it.seq.len - span) // stride + 1
    out = np.zeros((n, it.N), dtype=np.uint64)
    if n:
        d = it.ctx.alloc(out.nbytes)
        res = _capi.Result()
        try:
            view = it._view(0, it.seq.len)
            rc = it.ctx.check(it.ctx.lib.kmers_minimizers(it.ctx.handle, C.byref(view), it.K, W, stride, it.alphabet.bits,
                                                           mode, d, _capi.MEM_DEVICE, C.byref(res)), "kmers_minimizers")
            if rc == _capi.E_ENCODE:
                _raise_encode(it.alphabet, it.seq, res)
            it.ctx.d2h(out, d)
        finally:
            it.ctx.free(d)
    return KmerArray(it.alphabet, it.K, out)


def composition(it):
    """Kmer composition counts (docs/src/composition.md:28-39): counts[as_integer(kmer)] over
    FwKmers{DNA/RNAAlphabet{2},K}(seq), 4^K uint32 counters, one fused pass."""
    if not isinstance(it, FwKmers) or it.alphabet.bits != 2:
        raise UnsupportedError("composition(FwKmers over a 2-bit kmer alphabet)")
    counts = np.zeros(4 ** it.K, dtype=np.uint32)
    d = it.ctx.alloc(counts.nbytes)
    res = _capi.Result()
    try:
        view = it._view(0, it.seq.len)
        rc = it.ctx.check(it.ctx.lib.kmers_composition(it.ctx.handle, C.byref(view), it.K, d, _capi.MEM_DEVICE,
                                                        C.byref(res)), "kmers_composition")
        if rc == _capi.E_ENCODE:
            _raise_encode(it.alphabet, it.seq, res)
        it.ctx.d2h(counts, d)
    finally:
        it.ctx.free(d)
    return counts


# --------------------------------------------------------------------------------------------
# element-wise functions (Kmer or KmerArray)
def _batch(x):
    if isinstance(x, Kmer):
        return KmerArray(x.alphabet, x.K, np.array([x.data], dtype=np.uint64).reshape(1, -1)), True
    return x, False


def fx_hash(x, h=0, ctx=None):
    """fx_hash(x::Kmer, h::UInt) (src/kmer.jl:255-261) for one Kmer or a KmerArray."""
    ctx = ctx or default_context()
    if isinstance(x, Kmer) and x.K == 0:  # fx_hash of a 0-mer folds nothing: the library returns the seed
        out = np.zeros(1, dtype=np.uint64)
        ctx.check(ctx.lib.kmers_fx_hash(ctx.handle, out.ctypes.data_as(C.c_void_p), 0, 1, h & MASK64,
                                        out.ctypes.data_as(C.c_void_p), _capi.MEM_HOST), "kmers_fx_hash")
        return int(out[0])
    arr, single = _batch(x)
    n = len(arr)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    words = arr.words if arr.N else np.zeros(1, dtype=np.uint64)
    ctx.check(ctx.lib.kmers_fx_hash(ctx.handle, words.ctypes.data_as(C.c_void_p), arr.N, n, h & MASK64,
                                    out.ctypes.data_as(C.c_void_p), _capi.MEM_HOST), "kmers_fx_hash")
    return int(out[0]) if single else out[:n]


def _transform(op, x, ctx=None):
    ctx = ctx or default_context()
    arr, single = _batch(x)
    n = len(arr)
    width = 1 if op in (_capi.OP_ISCANONICAL, _capi.OP_COUNT_GC) else arr.N
    out = np.zeros((max(n, 1), width), dtype=np.uint64)
    if n:
        ctx.check(ctx.lib.kmers_transform(ctx.handle, op, arr.words.ctypes.data_as(C.c_void_p), arr.K,
                                          arr.alphabet.bits, n, out.ctypes.data_as(C.c_void_p),
                                          _capi.MEM_HOST), "kmers_transform")
    if op == _capi.OP_ISCANONICAL:
        return bool(out[0, 0]) if single else out[:n, 0].astype(bool)
    if op == _capi.OP_COUNT_GC:
        return int(out[0, 0]) if single else out[:n, 0].astype(np.int64)
    if op == _capi.OP_TO_LONGSEQ:
        if single:
            return LongSequence(arr.alphabet, out[0], arr.K)
        return out[:n]
    res = KmerArray(arr.alphabet, arr.K, out[:n])
    return res[0] if single else res


def reverse(x, ctx=None):
    return _transform(_capi.OP_REVERSE, x, ctx)             # transformations.jl:1-10


def complement(x, ctx=None):
    return _transform(_capi.OP_COMPLEMENT, x, ctx)          # transformations.jl:14-25


def reverse_complement(x, ctx=None):
    return _transform(_capi.OP_REVCOMP, x, ctx)             # transformations.jl:32-34


def canonical(x, ctx=None):
    return _transform(_capi.OP_CANONICAL, x, ctx)           # transformations.jl:36-39


def iscanonical(x, ctx=None):
    return _transform(_capi.OP_ISCANONICAL, x, ctx)         # transformations.jl:41


def count_gc(x, ctx=None):
    return _transform(_capi.OP_COUNT_GC, x, ctx)           # count(isGC, kmer), counting.jl:1-8


def to_longsequence(x, ctx=None):
    return _transform(_capi.OP_TO_LONGSEQ, x, ctx)         # LongSequence{A}(kmer), construction.jl:289-324


def as_integer(x, ctx=None):
    """as_integer(x::Kmer) (src/kmer.jl:305-326): the on-wire integer form (<= 128 bits).  A KmerArray
    is exported on the device: a uint64 array (<= 64 coding bits) or an (n, 2) array of little-endian
    u128 halves (low, high)."""
    if x.K * x.alphabet.bits > 128:
        raise KmersError("Must have at most 128 bits in encoding")
    if isinstance(x, KmerArray):
        ctx = ctx or default_context()
        n = len(x)
        out = np.zeros((max(n, 1), x.N), dtype=np.uint64)
        if n:
            ctx.check(ctx.lib.kmers_transform(ctx.handle, _capi.OP_AS_INTEGER, x.words.ctypes.data_as(C.c_void_p), x.K,
                                              x.alphabet.bits, n, out.ctypes.data_as(C.c_void_p), _capi.MEM_HOST),
                      "kmers_transform")
        return out[:n, 0] if x.N == 1 else out[:n]
    v = 0
    for w in x.data:
        v = (v << 64) | w
    return v


def from_integer(alphabet, K, u):
    """from_integer(T, u) (src/kmer.jl:361-384): only the low K*bits bits of u are used."""
    bits = K * alphabet.bits
    if bits > 128:
        raise KmersError("Kmer type must contain at most 128 bits")
    if isinstance(u, np.ndarray):  # batch: uint64[n] or uint64[n, 2] (little-endian u128 halves)
        ctx = default_context()
        N = n_coding_elements(K, alphabet.bits)
        u = np.ascontiguousarray(u, dtype=np.uint64).reshape(-1, N)
        out = np.zeros((max(len(u), 1), N), dtype=np.uint64)
        if len(u):
            ctx.check(ctx.lib.kmers_transform(ctx.handle, _capi.OP_FROM_INTEGER, u.ctypes.data_as(C.c_void_p), K,
                                              alphabet.bits, len(u), out.ctypes.data_as(C.c_void_p), _capi.MEM_HOST),
                      "kmers_transform")
        return KmerArray(alphabet, K, out[:len(u)])
    u &= (1 << bits) - 1 if bits else 0
    N = n_coding_elements(K, alphabet.bits)
    return Kmer(alphabet, K, tuple((u >> (64 * (N - 1 - i))) & MASK64 for i in range(N)))
