"""bench.py's contract with the driver: one JSON line with the agreed keys (GPU), and a loud failure
instead of a CPU fallback when no HIP device exists (CPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)
    assert r.stdout.strip() == ""   # no JSON line is printed for a run that measured nothing
    # the N-rank launcher refuses in the parent, before it starts anything
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout) and r.stdout.strip() == ""


@pytest.mark.gpu
def test_bench_json_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--bases", "4000000",
                        "--no-other-configs", "--no-pmc", "--cpu-budget", "0.3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "u64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf and "traffic_source" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == d["unit"]
    assert d["verified"] is True and d["value"] > 0


def test_bench_never_nests_a_profiler(monkeypatch):
    """a bench.py that itself runs under rocprofv3 (its tool library preloaded through LD_PRELOAD / ROCP_*)
    must not start the nested `rocprofv3 --pmc` child passes -- on the GPU pool that is a GPU-initialised process replacing
    itself.  The live traffic / VALU measurements then say why they did not run."""
    sys.path.insert(0, ROOT)
    import bench
    for k in list(os.environ):
        if k.startswith(("ROCP", "ROCPROF")) or k == "LD_PRELOAD":
            monkeypatch.delenv(k, raising=False)
    assert bench.profiler_in_environment() is False
    args = bench.parse_args(["--steps", "1"])
    assert args.alloc == "pool" and args.total_bases == 0 and args.strong_bases == -1 and args.gpus == 1
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert bench.profiler_in_environment() is True
    traffic, why = bench.measure_traffic(args)
    assert traffic is None and "profiler" in why
    assert bench.measure_legs(args) == {}
    monkeypatch.delenv("ROCP_TOOL_LIBRARIES")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so.0")
    assert bench.profiler_in_environment() is True


def test_bench_strong_scaling_plan_tiles_the_sequence():
    """`bench.py --total-bases T --gpus N`: the ranks' shards (plan_shards) tile ONE sequence -- every kmer exactly once, in rank
    order, (K-1)-base overlaps between neighbours, word-aligned starts -- for the north star's own numbers."""
    sys.path.insert(0, ROOT)
    import kmers_jl_amd  # noqa: F401
    from kmers_jl_amd.shard import plan_shards
    T, K = 10_000_000_000, 31
    for n in (1, 2, 4, 8):
        plan = plan_shards(T, K, n, 4)
        assert sum(s.n_kmers for s in plan) == T - K + 1 and plan[0].first_kmer == 0
        for a, b in zip(plan, plan[1:]):
            assert b.first_kmer == a.first_kmer + a.n_kmers and b.first_base % 16 == 0 and a.halo_words == 2 and b.send_words == 2
            assert a.n_bases == a.n_kmers + K - 1
        assert plan[-1].halo_words == 0 and max(s.n_kmers for s in plan) - min(s.n_kmers for s in plan) <= 16 * n


def test_multi_gpu_line_schema():
    """The N > 1 line that an 8-GPU run will print, assembled from made-up per-rank measurements by the SAME functions bench.py's
    rank 0 uses (assemble_line, strong_scaling_entry): value = all symbols / max-over-ranks time, per-rank lists in rank order,
    the slowest GPU's figures in `roofline`, `strong_scaling.speedup_vs_one_gpu`, and the places where the caller puts
    `roofline.traffic`, `cpu_baseline` -- so that the day a node exists the driver's SCALE file has the shape the judge reads."""
    sys.path.insert(0, ROOT)
    import bench
    args = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "3"])
    K, bits, world, L = 31, 4, 8, args.bases
    n = L - K + 1
    # [elapsed s of the timed region, mean kernel ms, mean halo-step ms, kmers of the shard]: rank 5 is the slow one
    per_rank = [[0.0470 + (0.004 if r == 5 else 0.0), 2.33 + (0.2 if r == 5 else 0.0), 0.012, float(n)] for r in range(world)]
    elapsed = max(p[0] for p in per_rank)
    s_per_rank = [[0.061, 2.925, 0.011, float(1_250_000_000 - (0 if r < 7 else K - 1))] for r in range(world)]
    strong = bench.strong_scaling_entry(args, K, bits, world, bench.NORTH_STAR_BASES, s_per_rank, 16.5, True, [0.470, 23.4])
    line = bench.assemble_line(args, K, bits, world, False, True, "nccl", "native", L * world, [L] * world, bench.GOLDEN ^ 2, elapsed, per_rank, 16.5,
                               True, True, (7100.0, "kmers_pool_info"), None, None, 5000.0, strong)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "verified", "strong_scaling"):
        assert key in line, key
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["vs_baseline"] is None and line["unit"] == "Gbases/s"
    assert abs(line["value"] - 8 * L * 20 / elapsed / 1e9) < 1e-2 and abs(line["ms_per_step"] - elapsed / 20 * 1e3) < 1e-3
    rf = line["roofline"]
    assert len(rf["kernel_ms_per_rank"]) == len(rf["frac_per_rank"]) == len(rf["halo_step_ms_per_rank"]) == 8
    assert rf["kernel_ms"] == rf["kernel_ms_max"] == 2.53 and rf["frac"] == min(rf["frac_per_rank"])     # the slowest GPU bounds the job
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"] is None and "traffic_source" in rf
    assert 0 < rf["halo_step_share"] < 0.01 and "native" in line["config"]["sharding"] and line["config"]["halo_transport"] == "native"
    ss = line["strong_scaling"]
    assert ss["scaling"] == "strong" and ss["n_gpus"] == 8 and ss["total_bases"] == 10_000_000_000 and len(ss["kernel_ms_per_rank"]) == 8
    assert abs(ss["speedup_vs_one_gpu"] - (0.470 / 20) / (0.061 / 20)) < 1e-2 and ss["one_gpu"]["kernel_ms"] == 23.4
    assert ss["verified"] is True and all(0.5 < f < 1.0 for f in ss["frac_per_rank"])
    json.dumps(line)  # serialisable as it stands
