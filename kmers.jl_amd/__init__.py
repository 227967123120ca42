"""kmers_jl_amd -- MI355X (gfx950) implementation of the k-mer iteration hot path of
BioJulia/Kmers.jl behind the reference's iterator / fx_hash / canonical API.

Layout: csrc/ (hand-written HIP kernels + the C ABI of include/kmers_hip.h), _capi (ctypes
binding of that ABI), host (Python mirror of the reference interface), shard (contiguous
multi-GPU sharding with a (K-1)-base halo).  No CPU compute path exists in this package.

A process that also uses PyTorch must import torch FIRST: torch brings its own copies of the HIP / HSA runtime libraries, and
once libkmers_hip.so has loaded /opt/rocm's, torch's cannot initialise ("No HIP GPUs are available").  With torch imported
first both share torch's copies.  The package itself needs only ctypes and numpy.
"""
from . import _capi, build  # noqa: F401
from .host import *  # noqa: F401,F403
from .host import (Context, EncodeError, KmerArray, Kmer, KmersError, LongSequence,  # noqa: F401
                   UnsupportedError, default_context)

__version__ = "0.1.0"
