#!/usr/bin/env python3
"""Files what `tools/evidence.sh` measured on the GPU box (gpurun_out/<round>ev/) under profiles/<round>_* (KMERS_ROUND, default r06).
    python tools/evidence.py [<dir of box 1> [<dir of box 2> ...]]      (directories under gpurun_out/; default <round>ev)
ONE table, a column per box (round 6; round 5 committed a file set per box): the files below are those of the FIRST box, a further
box adds its columns to <round>_legs.md and leaves its two bench lines as <round>_box<n>_bench*.json.


  <round>_bench.json                 the driver's command, as printed
  <round>_bench_under_rocprof.json   the same program under rocprofv3 --kernel-trace --stats (its line) ...
  <round>_kernel_stats.csv           ... and rocprofv3's per-kernel summary of that run
  <round>_kernel_stats_<leg>.csv     one --kernel-trace --stats pass per leg (tools/leg.py): each leg is a row of its own file
  <round>_legs.md                    per leg: rocprofv3's average kernel duration -> fraction of 8 TB/s on the algorithmic bytes of
                                 SURVEY.md 8(d), next to the leg's own HIP-event median; HBM bytes per launch from the
                                 FETCH_SIZE / WRITE_SIZE passes (gfx950 correction as MI355X_MICROARCH.md prescribes)
  pmc_traffic.json               the headline launch's bytes (bench.py replays it only when its live passes fail, and says so)
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("KMERS_ROUND", "r06")
BOXES = [os.path.join(ROOT, "gpurun_out", d) for d in (sys.argv[1:] or [RND + "ev"])]
E = BOXES[0]
P = os.path.join(ROOT, "profiles")
L = 1_000_000_000
LEGS = {  # leg -> (label, kernel substring, algorithmic bytes per launch given the leg's printed line)
    "c2": ("C2 CanonicalDNAMers{31} + fx_hash, 1 Gbase LongDNA{4}", "stream_kernel<4, 2, 1, 1", lambda kept: 16.5 * (L - 30)),
    "n1": ("N1 north star: CanonicalDNAMers{31} + fx_hash, 10 Gbase LongDNA{4}, ONE launch", "stream_kernel<4, 2, 1, 1", lambda kept: 16.5 * (10 * L - 30)),
    "c3": ("C3 CanonicalDNAMers{31}, 1.25 Gbase LongDNA{2}", "stream_kernel<2, 2, 1, 1", lambda kept: 8.25 * (1_250_000_000 - 30)),
    "c4": ("C4 FwDNAMers{63} + reverse complements", "stream_kernel<4, 2, 2, 0", lambda kept: 32.5 * (L - 62)),
    "c5": ("C5 strict SpacedDNAMers{21,3}", "stream_kernel<4, 2, 1, 0, false", lambda kept: 0.5 * L + 8.0 * ((L - 21) // 3 + 1)),
    "f3": ("f3 FwKmers{DNAAlphabet{4},31}: two-word kmers of a 4-bit alphabet, one array", "stream_kernel<4, 4, 2, 0", lambda kept: 16.5 * (L - 30)),
    "u31": ("UnambiguousDNAMers{31}, p(N) = 0.04", "unambiguous_kernel<4, 1, 0", lambda kept: 0.5 * L + 16.0 * kept),
    "u21": ("C5 skip variant: UnambiguousDNAMers{21} on the stride-3 lattice, p(N) = 0.04", "unambiguous_kernel<4, 1, 0", lambda kept: 0.5 * L + 16.0 * kept),
    "xor": ("fused XOR-reduce of CanonicalDNAMers{31}", "run_kernel<4, 0", lambda kept: 0.0),
    "minhash": ("fused MinHash candidates, CanonicalDNAMers{16}, bottom 1000", "run_kernel<4, 1", lambda kept: 0.0),
    "f1": ("f1 CanonicalDNAMers{31} + fx_hash from ASCII text", "stream_kernel<8, 2, 1, 1", lambda kept: 17.0 * (L - 30)),
    "f4h": ("f4 fx_hash over an array of 1 G one-word kmers", "fx_hash_kernel", lambda kept: 16.0 * L),
    "f4r": ("f4 reverse_complement over an array of 1 G DNAKmer{31}", "transform_kernel", lambda kept: 16.0 * L),
    "c63h": ("CanonicalDNAMers{63} + fx_hash (two-word kmers + hashes)", "stream_kernel<4, 2, 2, 1", lambda kept: 24.5 * (L - 62)),
    "f127": ("FwDNAMers{127} + reverse complements (four-word kmers)", "stream_kernel<4, 2, 4, 0", lambda kept: 64.5 * (L - 126)),
    "comp8": ("fused composition counts of FwDNAMers{8}", "composition_kernel", lambda kept: 0.0),
    "batch": ("kmers_batch: 8 M reads x 125 bases, CanonicalDNAMers{31} + fx_hash per read (element kernel only)", "ragged_kernel", lambda kept: 16.0 * 8_000_000 * 95 + 0.5 * 1e9),
    "batch_ascii": ("kmers_batch: the same reads from ASCII text (element kernel only)", "ragged_kernel", lambda kept: 16.0 * 8_000_000 * 95 + 1.0 * 1e9),
    "batch_n10": ("kmers_batch: an N in 10 % of the reads, KMERS_BATCH_SKIP, 4-bit pool (element kernel only)", "ragged_kernel", lambda kept: 16.0 * 8_000_000 * 95 + 0.5 * 1e9),
    "batch_ascii_n10": ("kmers_batch: an N in 10 % of the reads, KMERS_BATCH_SKIP, from text (element kernel only)", "ragged_kernel", lambda kept: 16.0 * 8_000_000 * 95 + 1.0 * 1e9),
    "batch_ragged": ("kmers_batch: 6.67 M reads of 50-250 bases, 4-bit pool (element kernel only)", "ragged_kernel", lambda kept: ragged_bytes(0.5)),
    "batch_ascii_ragged": ("kmers_batch: 6.67 M reads of 50-250 bases from text (element kernel only)", "ragged_kernel", lambda kept: ragged_bytes(1.0)),
}


def ragged_bytes(per_base):
    """algorithmic bytes of tools/batch_once.py --ragged --reads 6670000 (its lengths: numpy's default_rng(1), uniform 50..250)"""
    import numpy as np
    lengths = np.random.default_rng(1).integers(50, 251, 6_670_000)
    return 16.0 * float((lengths - 30).sum()) + per_base * float(lengths.sum())


def copy(src, dst):
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst))
        return True
    print("missing", src)
    return False


def first_json_line(path):
    try:
        for l in open(path):
            if l.startswith("{"):
                return json.loads(l)
    except OSError:
        pass
    return None


def main():
    os.makedirs(P, exist_ok=True)
    for name in ("bench.json", "bench_under_rocprof.json"):
        d = first_json_line(os.path.join(E, name))
        if d:
            json.dump(d, open(os.path.join(P, RND + "_" + name), "w"), indent=1)
    copy(os.path.join(E, "kernel_stats.csv"), RND + "_kernel_stats.csv")
    bench = first_json_line(os.path.join(E, "bench.json")) or {}
    nb = len(BOXES)
    rows = [f"# Round {RND[1:].lstrip('0')}: every leg in a rocprofv3 pass of its own (`tools/evidence.sh`, {nb} MI355X box{'es' if nb > 1 else ''}, outputs from the device's class pool)", "",
            f"`ms (rocprofv3)` = average duration of the leg's kernel in a `--kernel-trace --stats` pass over `tools/leg.py --leg <leg>` (warm-up launches",
            f"+ 20 timed ones; box 1's summaries are `profiles/{RND}_kernel_stats_<leg>.csv`); `ms (HIP events)` = the median the same process printed; fractions",
            "are of 8 TB/s on the algorithmic bytes of SURVEY.md 8(d).  HBM bytes: `FETCH_SIZE x 2` (gfx950: the counter reports half of a",
            "coalesced streaming read) `+ WRITE_SIZE`, separate passes over two bare launches (`leg.py --once`), box 1.", "",
            "| leg | kernel | " + " | ".join(f"box {b + 1}: ms (rocprofv3) / frac / ms (events) / frac" for b in range(nb)) + " | HBM bytes / algorithmic (box 1) |",
            "|---|---|" + "---|" * (nb + 1)]
    for leg, (label, sub, alg_of) in LEGS.items():
        cells, kname, traffic = [], "", ""
        for bi, box in enumerate(BOXES):
            stats = os.path.join(box, f"kernel_stats_{leg}.csv")
            if not os.path.exists(stats):
                cells.append("-")
                continue
            if bi == 0:
                copy(stats, f"{RND}_kernel_stats_{leg}.csv")
            kept, ev_ms, alg_printed = 0, None, None
            try:
                last = [l for l in open(os.path.join(box, f"stats_{leg}.txt")) if l.startswith(leg.split("_")[0]) or l.startswith("kmers_batch")][-1]
                m = re.search(r"kept=(\d+)", last)
                kept = int(m.group(1)) if m else 0
                m = re.search(r": ([0-9.]+) ms", last)
                ev_ms = float(m.group(1)) if m else None
            except (OSError, IndexError):
                pass
            alg = alg_of(kept)
            best = None
            for r in csv.DictReader(open(stats)):
                if sub in r["Name"] and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                    best = r
            if not best:
                cells.append("-")
                continue
            kname = kname or best["Name"][:70]
            ms = float(best["AverageNs"]) / 1e6
            fr = lambda t: f"{alg / t / 1e6 / 8000:.3f}" if alg and t else "-"
            cells.append(f"{ms:.4f} / {fr(ms)} / {ev_ms if ev_ms else '-'} / {fr(ev_ms)}")
            if bi == 0:
                vals = {}
                for c in ("FETCH_SIZE", "WRITE_SIZE"):
                    v = []
                    for f in glob.glob(os.path.join(box, f"pmc_{c}_{leg}", "**", "*counter_collection.csv"), recursive=True):
                        for r in csv.DictReader(open(f)):
                            if sub in r["Kernel_Name"] and r["Counter_Name"] == c:
                                v.append(float(r["Counter_Value"]))
                    if v:
                        vals[c] = sum(v) / len(v)
                if len(vals) == 2 and alg:
                    tot = vals["FETCH_SIZE"] * 2048 + vals["WRITE_SIZE"] * 1024
                    traffic = f"{tot / 1e9:.3f} GB / {alg / 1e9:.3f} GB = {tot / alg:.3f}"
        if kname:
            rows.append(f"| {label} | `{kname}` | " + " | ".join(cells) + f" | {traffic} |")
    for bi, box in enumerate(BOXES[1:], start=2):   # the further boxes' bench lines
        for name in ("bench.json", "bench_under_rocprof.json"):
            d = first_json_line(os.path.join(box, name))
            if d:
                json.dump(d, open(os.path.join(P, f"{RND}_box{bi}_{name}"), "w"), indent=1)
    if copy(os.path.join(E, "kernel_stats_minhash_batch.csv"), f"{RND}_kernel_stats_minhash_batch.csv"):
        try:
            lines = [l.strip() for l in open(os.path.join(E, "stats_minhash_batch.txt")) if "records" in l or "genomes" in l or "reads" in l]
            rows += ["", f"`kmers_minhash_batch` (`tools/sketch_batch_rate.py`; its kernels: `profiles/{RND}_kernel_stats_minhash_batch.csv`):", ""] + ["    " + l for l in lines[:8]]
        except OSError:
            pass
    # the headline launch by allocator
    sweep = []
    for bi, box in enumerate(BOXES):
      for f in sorted(glob.glob(os.path.join(box, "alloc_*.json"))):
        d = first_json_line(f)
        if d:
            cfgd = d.get("config", {})
            sweep.append(f"| box {bi + 1}: {cfgd.get('alloc', '?')} | {d['roofline']['kernel_ms']} | {d['roofline']['frac']} | {d['value']} | "
                         f"{(cfgd.get('pool') or {}).get('held_GB', '') or ''} |")
    if sweep:
        rows += ["", "## The headline launch (C2, 1 Gbase) by where its arrays come from (`bench.py --alloc ...`, same box, one process each)", "",
                 "| allocator | kernel ms | frac of 8 TB/s | Gbases/s | pool: GB held |", "|---|---|---|---|---|"] + sweep
    open(os.path.join(P, RND + "_legs.md"), "w").write("\n".join(rows) + "\n")
    rf = bench.get("roofline", {})
    if rf.get("traffic") and "measured in this run" in rf.get("traffic_source", ""):
        json.dump({"bases": bench["config"].get("bases_per_gpu"), "k": bench["config"]["k"], "src_bits": bench["config"]["src_bits"],
                   "traffic_bytes_per_launch": rf["traffic"], "source": f"profiles/{RND}_bench.json: " + rf["traffic_source"]},
                  open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    print("\n".join(rows))


if __name__ == "__main__":
    main()
