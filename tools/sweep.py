#!/usr/bin/env python3
"""Interleaved A/B sweep of launch tunables and library variants for the canonical+hash kernel
(cdna_hip_programming.md rule 24: variants x rounds in ONE process, report median and min).

    python tools/sweep.py --bases 1000000000 --tiles 1024,2048,4096,8192 --grids 0,2048,4096 \
        [--libs kmers.jl_amd/csrc/libkmers_hip.so,/path/variant.so] [--no-hash] [--src-bits 4]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import kmers_jl_amd as km  # noqa: E402

cap = km._capi


def load(path):
    lib = C.CDLL(path)
    for name, (res, args) in cap.SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bases", type=int, default=1_000_000_000)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--src-bits", type=int, default=4)
    ap.add_argument("--tiles", default="4096")
    ap.add_argument("--grids", default="0")
    ap.add_argument("--libs", default=cap.library_path())
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--no-hash", action="store_true")
    ap.add_argument("--prefetch", default="0", help="prefetch distances in tiles")
    ap.add_argument("--b-offsets", default="0", help="byte offsets of the second output array inside its allocation")
    ap.add_argument("--mode", default="canonical", choices=["canonical", "fw", "fwrc", "spaced"])
    ap.add_argument("--stride", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    K, bits, L = a.k, a.src_bits, a.bases
    n = L - K + 1 if a.mode != "spaced" else (L - K) // a.stride + 1
    N = (2 * K + 63) // 64
    nw = (L * bits + 63) // 64
    variants = []
    for path in a.libs.split(","):
        lib = load(path)
        h = C.c_void_p()
        assert lib.kmers_ctx_create(0, None, C.byref(h)) == 0
        for t in a.tiles.split(","):
            for g in a.grids.split(","):
                for off in a.b_offsets.split(","):
                    for pf in a.prefetch.split(","):
                        variants.append((os.path.basename(path), lib, h, int(t), int(g), int(off), int(pf)))
    lib0, h0 = variants[0][1], variants[0][2]
    buf = torch.zeros(nw + 2, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # the fill runs on torch's stream, the generator on the library's
    assert lib0.kmers_synth_dna(h0, 12345, 0, nw, bits, 0, buf.data_ptr()) == 0
    out_a = torch.empty(n * N, dtype=torch.int64, device=dev)
    max_off = max(int(o) for o in a.b_offsets.split(","))
    out_b = None if (a.no_hash or a.mode in ("fw", "spaced")) else torch.empty(n * (N if a.mode == "fwrc" else 1) + max_off // 8 + 2, dtype=torch.int64, device=dev)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    bpk = bits / 8 + 8 * N + (0 if out_b is None else (8 * N if a.mode == "fwrc" else 8))
    if a.mode == "spaced":
        bpk = bits / 8 * a.stride + 8 * N
    times = {i: [] for i in range(len(variants))}
    for rnd in range(a.rounds + 1):
        for i, (name, lib, h, t, g, off, pf) in enumerate(variants):
            pb = out_b.data_ptr() + off if out_b is not None else None
            lib.kmers_ctx_set_param(h, cap.PARAM_TILE_KMERS, t)
            lib.kmers_ctx_set_param(h, cap.PARAM_MAX_GRID, g)
            stream = torch.cuda.ExternalStream(lib.kmers_ctx_stream(h), device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(stream)
            if a.mode == "spaced":
                rc = lib.kmers_spaced(h, C.byref(seq), K, a.stride, 2, out_a.data_ptr(), cap.MEM_DEVICE | cap.ASYNC, C.byref(res))
            elif a.mode == "canonical":
                rc = lib.kmers_canonical(h, C.byref(seq), K, 2, out_a.data_ptr(), pb, 0, cap.MEM_DEVICE | cap.ASYNC, C.byref(res))
            else:
                rc = lib.kmers_fw(h, C.byref(seq), K, 2, out_a.data_ptr(), pb, cap.MEM_DEVICE | cap.ASYNC, C.byref(res))
            e1.record(stream)
            assert rc == 0
            torch.cuda.synchronize()
            if rnd:
                times[i].append(e0.elapsed_time(e1))
    print(f"# mode={a.mode} K={K} src_bits={bits} bases={L} bytes/kmer={bpk} rounds={a.rounds}")
    print(f"{'lib':28s} {'tile':>6s} {'grid':>8s} {'b_off':>9s} {'pf':>6s} {'med ms':>9s} {'min ms':>9s} {'GB/s(med)':>10s} {'frac8T':>7s}")
    for i, (name, lib, h, t, g, off, pf) in enumerate(variants):
        med, mn = float(np.median(times[i])), float(np.min(times[i]))
        gbs = bpk * n / (med * 1e-3) / 1e9
        print(f"{name:28s} {t:6d} {g:8d} {off:9d} {pf:6d} {med:9.4f} {mn:9.4f} {gbs:10.1f} {gbs / 8000:7.4f}")


if __name__ == "__main__":
    main()
