#!/usr/bin/env python3
"""Builds a tuning variant of libkmers_hip.so with extra compiler flags: tools/libkmers_<name>.so (git-ignored; it travels
to the GPU box with gpurun).  Select it with KMERS_HIP_LIB=... or tools/sweep.py --libs.

    python tools/build_variant.py wp -DKMERS_WAVE_PRIVATE
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmers_jl_amd  # noqa: E402

b = kmers_jl_amd.build
name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", f"libkmers_{name}.so")
rocm_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(b.hipcc()))), "lib")
cmd = [b.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-array-bounds", "-o", out] + flags + \
      [os.path.join(b.CSRC, s) for s in b.SOURCES] + [f"-L{rocm_lib}", "-lrccl"]
subprocess.run(cmd, check=True, cwd=b.CSRC)
print(out)
