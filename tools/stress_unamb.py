#!/usr/bin/env python3
"""Stress of the single-pass UnambiguousKmers kernel over any range of extra seeds: the cases of
tests/test_gpu_fuzz.py::test_unambiguous_geometries (random lengths, K, stride lattices, ambiguity patterns, tile sizes and grid
caps; device outputs compared element by element with the oracle).    python tools/stress_unamb.py [first_seed] [n]"""
import os
import sys
import time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch  # noqa: F401  (before the library: tests/test_gpu_pool.py says why)
import kmers_jl_amd as km
from oracle import pyoracle
import test_gpu_fuzz as tf
orc = pyoracle.get()
ctx = km.Context(0)
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
t0 = time.time()
for seed in range(first, first + count):
    tf.test_unambiguous_geometries(km, ctx, orc, seed)
    if (seed - first) % 20 == 19:
        print(f"seed {seed}: ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} cases from seed {first}: all equal to the oracle")
