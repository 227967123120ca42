#!/usr/bin/env python3
"""Checks the oracle (and the naive packers) against vectors produced by the REAL reference with
julia/make_golden.jl.  Julia is absent from the build image, so nobody has run this pairing there;
it is the tool that turns "LongSequence word order: parity unpinned by a known-answer test"
(tests/golden/README.md) into a pinned fact on any host that has Julia.

    julia --project=/path/to/Kmers.jl julia/make_golden.jl > /tmp/kats_from_julia.json
    python tests/golden/compare_with_julia.py /tmp/kats_from_julia.json

`--self-test` builds the same JSON from tests/naive.py instead (checks this script and that the
oracle and the naive model agree on exactly the vectors the Julia script emits)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import naive  # noqa: E402
from oracle import pyoracle  # noqa: E402

TEXT = "TTGCTAGGGATTCGAGGATCCTCTAGAGCGCGGCACGATC"
AMB = "TAGCWSAGACYWNACGCNACG--"


def hx(ws):
    return [f"0x{int(w):016x}" for w in ws]


def self_vectors():
    out = {"longseq_words": {"text": TEXT, "dna2": hx(naive.longseq_words(TEXT, 2)), "dna4": hx(naive.longseq_words(TEXT, 4))},
           "kmers": []}
    for K in (1, 7, 21, 31, 32, 33, 40):
        for name, bits in (("dna2", 2), ("dna4", 4)):
            t = TEXT[:K]
            rc = naive.revcomp_text(t)
            w, r = naive.kmer_words(t, bits), naive.kmer_words(rc, bits)
            out["kmers"].append({"alphabet": name, "text": t, "words": hx(w), "fx_hash": f"0x{naive.fx_hash(w):016x}",
                                 "revcomp": hx(r), "canonical": hx(min(w, r))})
    out["canonical31_first3"] = [hx(w) for w in naive.canonical(TEXT, 31, 2)[:3]]
    out["fwrv33_first2"] = [[hx(a), hx(b)] for a, b in naive.fwrv(TEXT, 33, 2)[:2]]
    out["unambiguous3"] = {"seq": AMB, "items": [[hx(w), i] for w, i in naive.unambiguous(AMB, 3)]}
    return out


def ints(hexes):
    return tuple(int(h, 16) for h in hexes)


def compare(v):
    orc = pyoracle.get()
    bad = []

    def check(what, got, exp):
        if got != exp:
            bad.append(f"{what}: oracle {got} != reference {exp}")
    text = v["longseq_words"]["text"]
    for name, bits in (("dna2", 2), ("dna4", 4)):
        check(f"LongSequence word order {name}", tuple(int(x) for x in naive.longseq_words(text, bits)),
              ints(v["longseq_words"][name]))
    for c in v["kmers"]:
        bits = 2 if c["alphabet"] == "dna2" else 4
        K = len(c["text"])
        seq = naive.longseq_words(c["text"], bits)
        w = orc.kmer_from_longseq(seq, K, K, bits)
        check(f"kmer layout {c['alphabet']} K={K}", w, ints(c["words"]))
        check(f"fx_hash {c['alphabet']} K={K}", orc.fx_hash(w), int(c["fx_hash"], 16))
        check(f"reverse_complement {c['alphabet']} K={K}", orc.reverse_complement(w, K, bits), ints(c["revcomp"]))
        check(f"canonical {c['alphabet']} K={K}", orc.canonical_kmer(w, K, bits), ints(c["canonical"]))
    seq4 = naive.longseq_words(text, 4)
    km, _, _ = orc.canonical(seq4, len(text), 4, 2, 31)
    check("CanonicalDNAMers{31} first 3", [tuple(int(x) for x in r) for r in km[:3]], [ints(h) for h in v["canonical31_first3"]])
    fw, rv, _ = orc.fwrv(seq4, len(text), 4, 2, 33)
    check("FwRvIterator{33} first 2", [(tuple(int(x) for x in a), tuple(int(x) for x in b)) for a, b in zip(fw[:2], rv[:2])],
          [(ints(a), ints(b)) for a, b in v["fwrv33_first2"]])
    amb = v["unambiguous3"]["seq"]
    km, st, _ = orc.unambiguous(naive.longseq_words(amb, 4), len(amb), 4, 3)
    check("UnambiguousDNAMers{3}", [(tuple(int(x) for x in a), int(i)) for a, i in zip(km, st)],
          [(ints(w), i) for w, i in v["unambiguous3"]["items"]])
    return bad


def main():
    v = self_vectors() if sys.argv[1:] == ["--self-test"] else json.load(open(sys.argv[1]))
    bad = compare(v)
    for b in bad:
        print("MISMATCH", b)
    print("all vectors agree" if not bad else f"{len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
