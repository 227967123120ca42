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
    assert args.alloc == "arena" and args.total_bases == 0 and args.strong_bases == -1 and args.gpus == 1
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
