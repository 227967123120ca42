#!/bin/bash
# Everything committed under profiles/<round>_* (KMERS_ROUND, default r06) comes from this script, run on the GPU box from the repo root:
#   gpurun --timeout 1700 -- 'bash tools/evidence.sh'      then, here:   python tools/evidence.py
#   1. the driver's command (bench.py with its own PMC child passes)                          -> bench.json
#   2. the same program under rocprofv3 --kernel-trace --stats, headline leg only (--no-other-configs: the other legs launch the
#      same kernel at other sizes and placements and would share its row)                      -> kernel_stats.csv
#   3. one --kernel-trace --stats pass PER LEG (tools/leg.py: one leg, one fresh process)      -> kernel_stats_<leg>.csv
#   4. one FETCH_SIZE and one WRITE_SIZE pass per leg (separate passes: TCC slots)             -> pmc_<counter>_<leg>/
#   5. kmers_batch under the same two kinds of pass (4-bit pool, text, an N in a tenth of the reads, ragged lengths); the headline
#      launch by allocator (pool / plain)
set -u
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
RND="${KMERS_ROUND:-r06}"
E="$ROOT/gpurun_out/${RND}ev"
rm -rf "$E"; mkdir -p "$E"
cd "$ROOT"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$E/bench.json" 2> "$E/bench.err"; echo "bench rc $?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$E/trace" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc > "$E/bench_under_rocprof.json" 2> "$E/trace.err"; echo "rocprof rc $?"
find "$E/trace" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats.csv"
for leg in c2 n1 c3 c4 c5 f1 f3 f4h f4r c63h f127 u31 u21 xor minhash comp8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$E/stats_$leg" -- python3 "$ROOT/tools/leg.py" --leg $leg --alloc pool --reps 20 > "$E/stats_$leg.txt" 2>&1
  find "$E/stats_$leg" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats_$leg.csv"
  rm -rf "$E/stats_$leg"
done
for leg in c2 n1 c3 c4 c5 f1 f3 f4h u31 u21; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$E/pmc_${c}_$leg" -- python3 "$ROOT/tools/leg.py" --leg $leg --alloc pool --once > "$E/pmc_${c}_$leg.txt" 2>&1
  done
done
# kmers_batch (8 M reads x 125 bases): the whole call's kernels, and the element kernel's bytes
rocprofv3 --kernel-trace --stats --output-format csv -d "$E/stats_batch" -- python3 "$ROOT/tools/batch_once.py" --reps 20 > "$E/stats_batch.txt" 2>&1
find "$E/stats_batch" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats_batch.csv"
rm -rf "$E/stats_batch"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$E/pmc_${c}_batch" -- python3 "$ROOT/tools/batch_once.py" --reps 2 > "$E/pmc_${c}_batch.txt" 2>&1
done
# ... on the reads people have (round 6): from text, with an N in a tenth of the reads (KMERS_BATCH_SKIP), lengths 50-250
for v in "ascii:--src 8" "ascii_n10:--src 8 --n-share 0.1 --skip" "n10:--src 4 --n-share 0.1 --skip" "ascii_ragged:--src 8 --ragged --reads 6670000" "ragged:--src 4 --ragged --reads 6670000"; do
  tag=${v%%:*}; opt=${v#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$E/stats_batch_$tag" -- python3 "$ROOT/tools/batch_once.py" $opt --reps 20 > "$E/stats_batch_$tag.txt" 2>&1
  find "$E/stats_batch_$tag" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats_batch_$tag.csv"
  rm -rf "$E/stats_batch_$tag"
done
# kmers_minhash_batch (one sketch per record, three batch shapes): its kernels in one trace
rocprofv3 --kernel-trace --stats --output-format csv -d "$E/stats_mhb" -- python3 "$ROOT/tools/sketch_batch_rate.py" > "$E/stats_minhash_batch.txt" 2>&1
find "$E/stats_mhb" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$E/kernel_stats_minhash_batch.csv"
rm -rf "$E/stats_mhb"
# where the headline's arrays come from: the class pool (default), plain allocations
cd "$ROOT"
for mode in "--alloc pool" "--alloc plain"; do
  python3 bench.py --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pmc $mode 2>/dev/null > "$E/alloc_$(echo $mode | tr -d ' -').json"
done
cd /tmp
cd "$ROOT"
rm -rf "$E/trace"
head -c 3000 "$E/bench.json"; echo; tail -3 "$E/bench.err"
for f in "$E"/stats_*.txt; do tail -1 "$f"; done
