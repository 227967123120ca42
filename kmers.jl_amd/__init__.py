"""kmers_jl_amd -- MI355X (gfx950) implementation of the k-mer iteration hot path of
BioJulia/Kmers.jl behind the reference's iterator / fx_hash / canonical API.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/kmers_hip.h), _capi (ctypes
binding of that ABI), host (Python mirror of the reference interface), shard (contiguous
multi-GPU sharding with a (K-1)-base halo).  No CPU compute path exists in this package.
"""
from . import _capi, build  # noqa: F401
from .host import *  # noqa: F401,F403
from .host import (Context, EncodeError, KmerArray, Kmer, KmersError, LongSequence,  # noqa: F401
                   UnsupportedError, default_context)

__version__ = "0.1.0"
