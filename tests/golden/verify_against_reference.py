#!/usr/bin/env python3
"""Provenance check of tests/golden/kats.json against the reference's own text.

Every group of kats.json cites the file:line range of BioJulia/Kmers.jl it was taken from.  This script opens those
ranges under the reference checkout (default /root/reference; it exists in the build container only), and looks every
literal of the group up in them: sequences, expected kmers, hexadecimal values, indices.  It writes
tests/golden/provenance.json -- literal -> file:line where the reference holds it -- which is committed, so that the
CPU suite can check (without the reference) that no vector was added or edited without being looked up again.
It reads the reference's text and stores line NUMBERS only; nothing of the reference is copied.

    python tests/golden/verify_against_reference.py [--reference /root/reference] [--check]
"""
import argparse
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
META_KEYS = {"cite", "alphabet", "dst", "src", "iter", "scheme", "seq_alphabet", "note", "derived"}
# Values the reference does not print and the fixture states as consequences of what it does print:
#   bits / as_integer_bits   the width of the printed unsigned literal (4 bits per hex digit; src/kmer.jl:305-326)
#   err_pos / err_symbol     the first symbol of the cited sequence that is not one of A C G T/U: the reference's tests only
#                            say that the iteration throws (test/runtests.jl:691-694, :868-869), FwKmers.jl:24-25 names the symbol
DERIVED_KEYS = {"bits", "as_integer_bits", "err_pos", "err_symbol"}


def cited_ranges(cite):
    """'src/a.jl:16-20, :70-75; docs/b.md:19' -> [(path, lo, hi), ...]"""
    out, path = [], None
    for m in re.finditer(r"([A-Za-z0-9_./-]+\.(?:jl|md))?:(\d+)(?:-(\d+))?", cite):
        if m.group(1):
            path = m.group(1)
        lo = int(m.group(2))
        hi = int(m.group(3) or lo)
        out.append((path, lo, hi))
    return out


def literals(node, key=None):
    """Every value of a group that has to be findable in the reference: strings and integers outside the metadata keys."""
    if isinstance(node, dict):
        for k, v in node.items():
            if k not in META_KEYS and k not in DERIVED_KEYS:
                yield from literals(v, k)
    elif isinstance(node, list):
        for v in node:
            yield from literals(v, key)
    elif isinstance(node, str):
        if node != "":
            yield key, node
    elif isinstance(node, bool):
        return
    elif isinstance(node, int):
        yield key, node


def find(lit, lines, lo, hi, slack):
    """Line numbers in [lo - slack, hi + slack] that hold the literal (sequences: case-insensitive; integers: as a token)."""
    hits = []
    for no in range(max(1, lo - slack), min(len(lines), hi + slack) + 1):
        text = lines[no - 1]
        if isinstance(lit, int):
            if re.search(r"(?<![0-9A-Za-z_])%d(?![0-9A-Za-z_])" % lit, text):
                hits.append(no)
        elif lit.lower().startswith("0x"):
            if lit.lower() in text.lower().replace("_", ""):
                hits.append(no)
        elif lit.lower() in text.lower():
            hits.append(no)
        elif lit.lower().replace("u", "t") in text.lower().replace("u", "t"):
            # the reference writes this one as RNA (rna"UAG...", LongRNA{2}("...")) and the fixture as DNA text, or the other
            # way round: T and U share their encoding in the 2- and 4-bit alphabets (BioSymbols), the packed words are the same
            hits.append(no)
    return hits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--slack", type=int, default=6, help="lines either side of a cited range that still count")
    ap.add_argument("--check", action="store_true", help="compare with the committed provenance.json instead of writing it")
    args = ap.parse_args()
    kats = json.load(open(os.path.join(HERE, "kats.json")))
    files = {}
    report, missing = {}, []
    for group, body in kats.items():
        if group.startswith("_"):
            continue
        ranges = cited_ranges(body["cite"])
        assert ranges, (group, body["cite"])
        entry = {"cite": body["cite"], "found": {}, "derived": body.get("derived", {})}
        for key, lit in literals(body):
            name = f"{key}={lit}"
            if name in entry["found"] or name in entry["derived"]:
                continue
            where = []
            for path, lo, hi in ranges:
                if path not in files:
                    files[path] = open(os.path.join(args.reference, path), encoding="utf-8").read().split("\n")
                where += [f"{path}:{no}" for no in find(lit, files[path], lo, hi, args.slack)]
            if where:
                entry["found"][name] = where[:4]
            else:
                missing.append((group, name))
        report[group] = entry
    for group, name in missing:
        print(f"NOT FOUND in the cited lines: {group}: {name}")
    n = sum(len(e["found"]) for e in report.values())
    print(f"{n} literals found in the cited ranges, {len(missing)} missing, "
          f"{sum(len(e['derived']) for e in report.values())} declared as derived")
    path = os.path.join(HERE, "provenance.json")
    if args.check:
        old = json.load(open(path))
        if old != report:
            print("provenance.json is stale: run this script without --check")
            return 1
    elif not missing:
        json.dump(report, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", path)
    return 1 if missing else 0


if __name__ == "__main__":
    sys.exit(main())
