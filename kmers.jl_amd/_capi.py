"""ctypes binding of include/kmers_hip.h (the C ABI of libkmers_hip.so).

This is the same binding a Julia `@ccall` shim would make.  There is no
fallback: if the library is missing or no HIP device is usable, loading / context creation
raises.
"""
import ctypes as C
import os

from . import build as _build

OK, E_ENCODE, E_BADARG, E_HIP, E_NOMEM, E_UNSUPPORTED, E_CAPACITY, E_NCCL = range(8)
COMM_ID_BYTES = 128
MEM_HOST, MEM_DEVICE, ASYNC, OUT_TUPLES = 0, 1, 2, 4
BATCH_FW, BATCH_CANONICAL = 0, 1
ITER_FW, ITER_CANONICAL, ITER_SPACED, ITER_UNAMBIGUOUS = 0, 1, 2, 3
SPANS_DEVICE = 8
BATCH_SKIP = 16
(OP_REVERSE, OP_COMPLEMENT, OP_REVCOMP, OP_CANONICAL, OP_ISCANONICAL, OP_TO_LONGSEQ, OP_COUNT_GC, OP_AS_INTEGER,
 OP_FROM_INTEGER) = range(9)
PARAM_TILE_KMERS, PARAM_MAX_GRID, PARAM_STAMPS_PTR, PARAM_SKETCH_HOST_ONLY, PARAM_BATCH_PASSES, PARAM_SKETCH_BATCH_LDS, PARAM_SUBTILES, PARAM_SPLIT_ORDER, PARAM_BLOCK_THREADS = 1, 2, 3, 4, 5, 6, 7, 9, 10
PARAM_WIDE_NO_TILES, PARAM_HOST_CHUNKS = 11, 12
PARAM_POOL, PARAM_POOL_SEARCH_GIB, PARAM_POOL_MAX_GIB, PARAM_BATCH_DENSE, PARAM_POOL_CACHE = 14, 15, 16, 17, 18
POOL_MIN_BYTES, POOL_CLASSES, POOL_STATS = 128 << 20, 4, 12
ALLOC_DEFAULT, ALLOC_LONE_OUTPUT = 0, 1

STATUS_NAMES = {OK: "KMERS_OK", E_ENCODE: "KMERS_E_ENCODE", E_BADARG: "KMERS_E_BADARG",
                E_HIP: "KMERS_E_HIP", E_NOMEM: "KMERS_E_NOMEM",
                E_UNSUPPORTED: "KMERS_E_UNSUPPORTED", E_CAPACITY: "KMERS_E_CAPACITY", E_NCCL: "KMERS_E_NCCL"}


class Result(C.Structure):
    _fields_ = [("status", C.c_int32), ("err_enc", C.c_uint32), ("err_pos", C.c_uint64),
                ("n_out", C.c_uint64)]


class Span(C.Structure):
    _fields_ = [("first_base", C.c_uint64), ("n_bases", C.c_uint64)]


class ShardPlan(C.Structure):
    """kmers_shard of include/kmers_hip.h"""
    _fields_ = [("first_kmer", C.c_uint64), ("n_kmers", C.c_uint64), ("first_base", C.c_uint64), ("n_bases", C.c_uint64),
                ("first_word", C.c_uint64), ("n_own_words", C.c_uint64), ("halo_words", C.c_uint32), ("send_words", C.c_uint32)]


class Seq(C.Structure):
    _fields_ = [("words", C.c_void_p), ("n_bases", C.c_uint64), ("first_base", C.c_uint64),
                ("index_origin", C.c_uint64), ("src_bits", C.c_int32), ("alphabet", C.c_int32)]


# every symbol include/kmers_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_R = C.POINTER(Result)
_S = C.POINTER(Seq)
SYMBOLS = {
    "kmers_abi_version": (C.c_int, []),
    "kmers_ctx_create": (C.c_int, [C.c_int, _P, C.POINTER(_P)]),
    "kmers_ctx_destroy": (None, [_P]),
    "kmers_ctx_stream": (_P, [_P]),
    "kmers_last_error": (C.c_char_p, [_P]),
    "kmers_sync": (C.c_int, [_P, _R]),
    "kmers_ctx_set_param": (C.c_int, [_P, C.c_int, C.c_int64]),
    "kmers_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "kmers_dev_alloc_role": (C.c_int, [_P, C.c_size_t, C.c_int, C.POINTER(_P)]),
    "kmers_dev_free": (C.c_int, [_P, _P]),
    "kmers_last_launch_shape": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kmers_last_batch_pieces": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "kmers_pool_info": (C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
    "kmers_pool_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_size_t]),
    "kmers_pool_trim": (C.c_int, [_P, C.POINTER(C.c_size_t)]),
    "kmers_pool_layout": (C.c_int, [_P, _P, C.POINTER(C.c_size_t), _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "kmers_pool_selftest": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "kmers_placement_probe": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_double)]),
    "kmers_memcpy_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kmers_memcpy_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kmers_host_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "kmers_host_free": (C.c_int, [_P, _P]),
    "kmers_memcpy_h2d_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kmers_memcpy_d2h_async": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "kmers_words_per_kmer": (C.c_int, [C.c_int, C.c_int]),
    "kmers_count": (C.c_uint64, [C.c_uint64, C.c_int, C.c_int]),
    "kmers_supported": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "kmers_fw": (C.c_int, [_P, _S, C.c_int, C.c_int, _P, _P, C.c_int, _R]),
    "kmers_canonical": (C.c_int, [_P, _S, C.c_int, C.c_int, _P, _P, C.c_uint64, C.c_int, _R]),
    "kmers_spaced": (C.c_int, [_P, _S, C.c_int, C.c_int, C.c_int, _P, C.c_int, _R]),
    "kmers_unambiguous": (C.c_int, [_P, _S, C.c_int, C.c_int, _P, _P, C.c_uint64, C.c_int, _R]),
    "kmers_reduce_xor": (C.c_int, [_P, _S, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int, _R]),
    "kmers_reduce_xor_iter": (C.c_int, [_P, _S, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.c_int, _R]),
    "kmers_minhash": (C.c_int, [_P, _S, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, C.c_int, _R]),
    "kmers_minimizers": (C.c_int, [_P, _S, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, _R]),
    "kmers_composition": (C.c_int, [_P, _S, C.c_int, _P, C.c_int, _R]),
    "kmers_fx_hash": (C.c_int, [_P, _P, C.c_int, C.c_uint64, C.c_uint64, _P, C.c_int]),
    "kmers_transform": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_uint64, _P, C.c_int]),
    "kmers_batch": (C.c_int, [_P, _S, _P, C.c_uint64, C.c_int, C.c_int, C.c_int, _P, _P, C.c_uint64, _P, C.c_uint64, C.c_int, _R]),
    "kmers_batch_spaced": (C.c_int, [_P, _S, _P, C.c_uint64, C.c_int, C.c_uint64, C.c_int, _P, _P, C.c_uint64, C.c_int, _R]),
    "kmers_minhash_batch": (C.c_int, [_P, _S, _P, C.c_uint64, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, _P, C.c_int, _R]),
    "kmers_shard_plan": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.c_int, C.c_int, C.c_int, _P]),
    "kmers_comm_id": (C.c_int, [_P]),
    "kmers_comm_create": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "kmers_comm_destroy": (C.c_int, [_P, _P]),
    "kmers_comm_rank": (C.c_int, [_P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kmers_comm_sendrecv": (C.c_int, [_P, _P, _P, C.c_uint64, C.c_int, _P, C.c_uint64, C.c_int]),
    "kmers_halo_exchange": (C.c_int, [_P, _P, C.POINTER(ShardPlan), _P]),
    "kmers_first_error_allreduce": (C.c_int, [_P, _P, _R]),
    "kmers_offsets_allgather": (C.c_int, [_P, _P, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "kmers_synth_dna": (C.c_int, [_P, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, _P]),
}

_lib = None


def library_path():
    # KMERS_HIP_LIB selects another build of the SAME library (tuning variants, diagnostic builds)
    return os.environ.get("KMERS_HIP_LIB") or _build.LIB


def load():
    """Load libkmers_hip.so (must have been built: `python -m kmers_jl_amd.build` or
    __graft_entry__.build()).  Raises -- never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: the HIP extension has not been built (run __graft_entry__.build()). "
            "kmers_jl_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
