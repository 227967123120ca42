"""tests/golden/kats.json against its provenance record (tests/golden/provenance.json): every literal of every group was
looked up in the file:line range of the reference it cites, by tests/golden/verify_against_reference.py.  Without the
reference (the GPU box) the test checks that the record covers the fixture as it is now; with it (/root/reference, the
build container) it repeats the look-up."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
spec = importlib.util.spec_from_file_location("verify_against_reference", os.path.join(GOLDEN, "verify_against_reference.py"))
verify = importlib.util.module_from_spec(spec)
spec.loader.exec_module(verify)


def test_provenance_record_covers_every_literal():
    kats = json.load(open(os.path.join(GOLDEN, "kats.json")))
    record = json.load(open(os.path.join(GOLDEN, "provenance.json")))
    groups = [g for g in kats if not g.startswith("_")]
    assert sorted(groups) == sorted(record)
    n = 0
    for g in groups:
        assert record[g]["cite"] == kats[g]["cite"]
        for key, lit in verify.literals(kats[g]):
            name = f"{key}={lit}"
            assert name in record[g]["found"] or name in record[g]["derived"], (g, name)
            for where in record[g]["found"].get(name, []):
                path, line = where.rsplit(":", 1)
                assert any(path == p and lo - 6 <= int(line) <= hi + 6 for p, lo, hi in verify.cited_ranges(kats[g]["cite"])), (g, name, where)
            n += 1
    assert n > 150


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference checkout exists in the build container only")
def test_lookup_repeats_against_the_reference_text():
    out = subprocess.run([sys.executable, os.path.join(GOLDEN, "verify_against_reference.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert " 0 missing" in out.stdout
