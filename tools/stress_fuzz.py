#!/usr/bin/env python3
"""One-off stress: the seeded fuzz tests of tests/test_gpu_fuzz.py with many more seeds."""
import os
import sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import json
import kmers_jl_amd as km
from oracle import pyoracle
import test_gpu_fuzz as tf
orc = pyoracle.get()
ctx = km.Context(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    tf.test_fuzz_iterators.__wrapped__(km, ctx, orc, seed) if hasattr(tf.test_fuzz_iterators, "__wrapped__") else tf.test_fuzz_iterators(km, ctx, orc, seed)
    if seed % 3 == 0:
        tf.test_fuzz_fused_consumers(km, ctx, orc, seed)
    tf.test_fuzz_batches(km, ctx, orc, seed)
    tf.test_fuzz_batches_of_reads_with_ambiguous_symbols(km, ctx, orc, seed)
    print("seed", seed, "ok", flush=True)
