#!/usr/bin/env python3
"""Writes tests/golden/synth_10k.json: the 10 kbase synthetic fixture of SURVEY.md section 8d.

The synthetic generator is this build's own (SplitMix64 stream, DESIGN.md section 5 -- Julia's RNG is
not reproducible here), so this file does not pin the reference; it pins the GENERATOR and the
end-to-end outputs on it, so that the host oracle, the device generator (kmers_synth_dna) and the
HIP iterators cannot drift apart silently.  Expected values come from the oracle, which is itself
pinned on the reference's known-answer vectors (kats.json).

    python tests/golden/make_synth_fixture.py      # rewrites synth_10k.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle  # noqa: E402

N_BASES = 10_000
SEED = 0x9E3779B97F4A7C15 ^ 2  # golden ratio constant xor config id (C2)


def hx(a):
    return [f"0x{int(x):016x}" for x in np.asarray(a, dtype=np.uint64).reshape(-1)]


def main():
    orc = pyoracle.get()
    out = {"_about": "Synthetic 10 kbase fixture (generator + end-to-end pin; see make_synth_fixture.py). "
                     "Not a reference vector: the generator is this build's own.",
           "seed": f"0x{SEED:016x}", "n_bases": N_BASES, "cases": []}
    for bits in (2, 4):
        n_words = (N_BASES * bits + 63) // 64
        words = orc.synth_words(SEED, 0, n_words + 1, bits)
        for K in (21, 31, 63):
            km, hs, res = orc.canonical(words, N_BASES, bits, 2, K, seed=0)
            assert res.status == 0
            fold = [int(np.bitwise_xor.reduce(km[:, j])) for j in range(km.shape[1])]
            out["cases"].append({
                "iter": "canonical", "src_bits": bits, "K": K, "n": int(len(km)),
                "source_words_first4": hx(words[:4]), "source_xor": hx([np.bitwise_xor.reduce(words[:n_words])])[0],
                "kmer_xor": hx(fold), "hash_xor": hx([np.bitwise_xor.reduce(hs)])[0],
                "first16": hx(km[:16]), "last16": hx(km[-16:]),
                "first16_hashes": hx(hs[:16]), "last16_hashes": hx(hs[-16:])})
    # C5 flavour: 4-bit source, each base N with p = 2621/65536 = 0.04 (test/utils.jl:22-24)
    words = orc.synth_words(SEED ^ 7, 0, N_BASES // 16 + 1, 4, 2621)
    km, st, res = orc.unambiguous(words, N_BASES, 4, 21)
    lattice = (st - 1) % 3 == 0
    _, sres = orc.spaced(words, N_BASES, 4, 2, 21, 3)
    out["cases"].append({
        "iter": "unambiguous", "src_bits": 4, "K": 21, "ambig_per_65536": 2621, "seed": f"0x{SEED ^ 7:016x}",
        "n": int(len(km)), "kmer_xor": hx([np.bitwise_xor.reduce(km[:, 0])])[0],
        "start_sum": int(st.sum()), "first16_starts": [int(x) for x in st[:16]],
        "lattice3_n": int(lattice.sum()), "lattice3_kmer_xor": hx([np.bitwise_xor.reduce(km[lattice, 0])])[0],
        "strict_spaced_error": {"pos": int(sres.err_pos), "enc": int(sres.err_enc)}})
    with open(os.path.join(HERE, "synth_10k.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
