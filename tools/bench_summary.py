#!/usr/bin/env python3
"""One line per leg of a bench.py JSON line (stdin or a file): fraction of 8 TB/s and milliseconds."""
import json
import sys

d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read())
rf = d["roofline"]
print("headline", rf["frac"], "plain allocations", (rf.get("plain_alloc") or {}).get("frac"), "|", d["config"].get("pool"))
for k, v in d.get("other_configs", {}).items():
    print(f"  {k[:64]:64s} {v.get('frac_of_8TBps')}  {v.get('kernel_ms', v.get('ms'))} ms")
