#!/usr/bin/env python3
"""Issue cost of gfx950 code by instruction class (profiles/r04_valu_rates.txt): v_and / v_or / v_xor / v_add / v_sub / v_mov / v_not /
right shifts over registers, inline constants or literals (and v_bitop3 over registers) cost a SIMD 2.24 cycles per wave64
instruction; every other vector instruction -- v_lshlrev_b32 and any of the above with an SGPR operand included -- 4.1; ds_read 8, ds_write 16,
ds_bpermute 24 cycles of the CU's LDS pipe.  Static: per kernel (or per label range) the number of instructions of each class
and their cycles -- loops count once, so compare straight-line code (an unrolled step, a device function in a test kernel).

  hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only -o x.s x.hip ; python3 tools/isa_cost.py x.s [kernel-substring] [--blocks]
"""
import re
import sys
from collections import Counter

FAST = {"v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
        "v_mov_b32", "v_not_b32"}   # measured at 2.24 (v_subrev assumed like v_sub); v_lshlrev_b32 is NOT: 4.1
FAST_IF_VGPR = {"v_bitop3_b32"}


def classify(op, args):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base.startswith("v_"):
        sgpr = re.search(r"(^|[ ,])(s\d+|s\[\d+:\d+\]|vcc|exec)", args.split(",", 1)[1] if "," in args else "")  # a scalar SOURCE operand
        if base in FAST and not op.endswith(("_e64", "_dpp", "_sdwa")) and not sgpr:
            return "fast"
        if base in FAST_IF_VGPR and not re.search(r"\bs\d|\bs\[|vcc|exec", args):
            return "fast"
        return "slow"
    if base.startswith("ds_"):
        if "bpermute" in base or "permute" in base:
            return "lds_perm"
        return "lds_write" if ("write" in base or "_add" in base or "_or" in base or "_min" in base or "_max" in base) else "lds_read"
    if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if base == "s_nop":
        return "nop"
    if base.startswith("s_"):
        return "salu"
    return "other"


COST = {"fast": 2.24, "slow": 4.1, "lds_read": 0, "lds_write": 0, "lds_perm": 0, "vmem": 0, "nop": 1.0, "salu": 0, "other": 0}
LDS = {"lds_read": 8.1, "lds_write": 16.1, "lds_perm": 24.0}


def main():
    path = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    blocks = "--blocks" in sys.argv
    kernels, name, lines = [], None, []
    for raw in open(path):
        line = raw.rstrip("\n")
        if name is None:
            if line.startswith("_Z") and line.rstrip().endswith(":") or (line.startswith("_Z") and ":" in line.split(";")[0]):
                name, lines = line.split(":")[0], []
        else:
            lines.append(line)
            if "s_endpgm" in line:
                kernels.append((name, lines))
                name = None
    for name, body in kernels:
        if pat not in name:
            continue
        total, per_block, label = Counter(), [], "entry"
        cur = Counter()
        for line in body:
            line = line.split(";")[0].strip()
            if not line:
                continue
            if line.endswith(":"):
                per_block.append((label, cur))
                label, cur = line[:-1], Counter()
                continue
            if line.startswith("."):
                continue
            parts = line.split(None, 1)
            c = classify(parts[0], parts[1] if len(parts) > 1 else "")
            cur[c] += 1
            total[c] += 1
        per_block.append((label, cur))

        def cycles(c):
            return sum(COST[k] * v for k, v in c.items()), sum(LDS.get(k, 0) * v for k, v in c.items())
        simd, lds = cycles(total)
        print(f"{name[:90]}\n   fast {total['fast']}  slow {total['slow']}  s_nop {total['nop']}  salu {total['salu']}  lds r/w/perm "
              f"{total['lds_read']}/{total['lds_write']}/{total['lds_perm']}  vmem {total['vmem']}   -> {simd:.0f} SIMD cycles + {lds:.0f} LDS-pipe cycles (static)")
        if blocks:
            for label, c in per_block:
                n = sum(c.values())
                if n >= 12:
                    s, l = cycles(c)
                    print(f"      {label:12s} {n:5d} instr: fast {c['fast']:4d} slow {c['slow']:4d} lds {c['lds_read']}/{c['lds_write']}/{c['lds_perm']}  {s:7.0f} + {l:5.0f}")


if __name__ == "__main__":
    main()
