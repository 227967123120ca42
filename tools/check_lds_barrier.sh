#!/bin/bash
# "the LDS-atomics bare s_barrier hazard was fixed empirically; the root cause is asserted, not shown in ISA".
# This compiles the device code twice -- as shipped, and with block_sync()'s explicit `s_waitcnt lgkmcnt(0)` removed
# (-DKMERS_NO_SETTLE) -- and lists every s_barrier that has an LDS operation issued since the last full lgkmcnt(0) wait.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="${1:-/tmp/kmers_isa}"
mkdir -p "$OUT"
cd "$ROOT/kmers.jl_amd/csrc"
: > "$OUT/settle.s"; : > "$OUT/nosettle.s"
for f in *_api.hip; do   # every translation unit of the library, concatenated
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$f" -o "$OUT/one.s" 2>/dev/null && cat "$OUT/one.s" >> "$OUT/settle.s"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -DKMERS_NO_SETTLE "$f" -o "$OUT/one.s" 2>/dev/null && cat "$OUT/one.s" >> "$OUT/nosettle.s"
done
python3 - "$OUT" <<'PY'
import re, sys
out = sys.argv[1]
for tag in ("settle", "nosettle"):
    s = open(f"{out}/{tag}.s").read()
    total = pending = 0
    where = []
    for f in re.split(r"\n(?=_ZN5kmers[^\n]*:\s+; @)", s):
        name = f.split(":", 1)[0]
        lines = [l.strip() for l in f.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        for i, l in enumerate(lines):
            if not l.startswith("s_barrier"):
                continue
            total += 1
            j, op = i - 1, None
            while j >= 0 and not lines[j].startswith("s_barrier"):
                if lines[j].startswith("s_waitcnt") and "lgkmcnt(0)" in lines[j]:
                    break
                if lines[j].startswith("ds_"):
                    op = lines[j]
                j -= 1
            if op:
                pending += 1
                where.append((name[:70], op))
    print(f"{tag:9s}: {total} s_barrier, {pending} with an LDS operation issued since the last s_waitcnt lgkmcnt(0)")
    for w in where:
        print("          ", *w)
PY
