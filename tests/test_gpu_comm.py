"""The communication of the sharded path behind the C ABI (include/kmers_hip.h: kmers_comm_*, kmers_halo_exchange,
kmers_first_error_allreduce, kmers_offsets_allgather) on a real RCCL communicator.  A 1-GPU box admits one rank per
communicator (RCCL refuses two ranks on one device), so the hardware legs here are: the communicator life cycle, a grouped
ncclSend/ncclRecv through kmers_comm_sendrecv (peer = self), the two reductions on one rank, and `bench.py --gpus 2`
end to end with two ranks sharing the device over gloo (both torch transports).  The N > 1 RCCL run itself is the driver's."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    return kmers_jl_amd


@pytest.fixture(scope="module")
def ctx(km):
    c = km.Context(0)
    yield c
    c.close()


def test_single_rank_communicator(km, ctx):
    from kmers_jl_amd.shard import NativeComm, plan_shards
    cap = km._capi
    ident = NativeComm.new_id(ctx.lib)
    assert len(ident) == cap.COMM_ID_BYTES
    comm = NativeComm.create(ctx, ident, 1, 0)
    r, n = C.c_int(-1), C.c_int(-1)
    assert ctx.lib.kmers_comm_rank(ctx.handle, comm.handle, C.byref(r), C.byref(n)) == 0 and (r.value, n.value) == (0, 1)
    # grouped ncclSend + ncclRecv on the context's stream (peer = this rank): words [0, 4) -> words [10, 14)
    host = np.arange(1, 17, dtype=np.uint64) * np.uint64(0x0123456789ABCDEF)
    d = ctx.alloc(host.nbytes)
    ctx.h2d(d, host)
    comm.sendrecv(d, 4, 0, d + 80, 4, 0)
    back = np.zeros_like(host)
    ctx.d2h(back, d)   # same stream: ordered after the exchange
    want = host.copy()
    want[10:14] = host[0:4]
    assert np.array_equal(back, want)
    # a one-shard plan has no neighbour: the exchange enqueues nothing and succeeds
    plan = plan_shards(10_000, 31, 1, 4)
    comm.halo_exchange(plan[0], d)
    # a two-shard plan on a one-rank communicator: rank 0 of 1 neither sends (rank 0) nor receives (last rank)
    comm.halo_exchange(plan_shards(10_000, 31, 2, 4)[0], d)
    ctx.d2h(back, d)
    assert np.array_equal(back, want)
    # the two reductions
    assert comm.first_error(0) == (0, 0, 0)
    assert comm.first_error(1, err_pos=123_456_789_012, err_enc=0xF) == (1, 123_456_789_012, 0xF)
    assert comm.first_error(1, err_pos=7, err_enc=ord("!")) == (1, 7, ord("!"))
    assert comm.output_offsets(0) == (0, 0)
    assert comm.output_offsets(987_654_321_000) == (0, 987_654_321_000)
    # bad arguments come back as statuses, never as aborts
    res = cap.Result(cap.E_HIP, 0, 0, 0)
    assert ctx.lib.kmers_first_error_allreduce(ctx.handle, comm.handle, C.byref(res)) == cap.E_BADARG
    assert ctx.lib.kmers_halo_exchange(ctx.handle, None, None, None) == cap.E_BADARG
    assert ctx.lib.kmers_comm_sendrecv(ctx.handle, comm.handle, None, 4, 0, None, 0, -1) == cap.E_BADARG
    ctx.free(d)
    comm.close()


def run_bench(extra_args, env_extra, timeout=900):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, capture_output=True, text=True, timeout=timeout, env=env)
    return r


@pytest.mark.parametrize("transport", ["allgather", "p2p"])
def test_bench_launches_two_ranks_by_itself(transport):
    """`python bench.py --gpus 2` exactly as the driver calls it (no torchrun around it): the parent starts the ranks.  Two
    ranks share the one device over gloo here, which exercises the launcher, the shard plan, the halo step and the
    verification of both shards; the line must report what really ran."""
    r = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--bases", "8000000", "--strong-bases", "12000029", "--no-other-configs",
                   "--cpu-budget", "0.2", "--no-pmc"],
                  {"KMERS_BENCH_BACKEND": "gloo", "KMERS_HALO_TRANSPORT": transport})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["scaling"] == "weak"
    assert d["config"]["backend"] == "gloo" and d["config"]["halo_transport"] == transport
    assert "gloo" in d["config"]["sharding"] and "RCCL" not in d["config"]["sharding"]
    assert d["value"] > 0 and d["roofline"]["kmers_per_launch"] > 0
    # the N > 1 line carries what the N = 1 line does: per-rank kernel times, the halo step's share, the CPU baseline
    rf = d["roofline"]
    assert len(rf["kernel_ms_per_rank"]) == 2 and rf["kernel_ms_min"] <= rf["kernel_ms"] <= rf["kernel_ms_max"] + 1e-9
    assert len(rf["halo_step_ms_per_rank"]) == 2 and 0 <= rf["halo_step_share"] < 1 and "traffic_source" in rf
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    # ... and the strong split of a fixed input measured in the same run, with the same input on one rank beside it
    st = d["strong_scaling"]
    assert st["scaling"] == "strong" and st["total_bases"] == 12000029 and st["n_gpus"] == 2 and st["verified"] is True
    assert len(st["kernel_ms_per_rank"]) == 2 and st["one_gpu"]["ms_per_step"] > 0 and st["speedup_vs_one_gpu"] > 0


def test_bench_strong_scaling_eight_ranks_share_the_device():
    """`python bench.py --gpus 8 --total-bases T`: ONE sequence of T symbols split over 8 ranks by kmers_shard_plan (the north
    star's split at a small size; the ranks share this box's one device over gloo).  Every rank verifies its shard against the
    oracle at both ends and by the all-element checks; the line says "strong" and names the split."""
    total = 40_000_003
    r = run_bench(["--gpus", "8", "--total-bases", str(total), "--steps", "2", "--warmup", "1", "--no-other-configs", "--no-cpu-baseline", "--no-pmc"],
                  {"KMERS_BENCH_BACKEND": "gloo"}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["verified"] is True
    assert d["config"]["total_bases"] == total and len(d["config"]["bases_per_gpu"]) == 8
    # the shards tile the sequence: every kmer exactly once, (K-1)-base overlaps between neighbours
    assert sum(b - 30 for b in d["config"]["bases_per_gpu"]) == total - 30
    assert "ONE sequence" in d["config"]["workload"] and "8 GPU" in d["config"]["workload"]
    assert len(d["roofline"]["kernel_ms_per_rank"]) == 8 and abs(d["value"] - total * 2 / (d["ms_per_step"] * 2e-3) / 1e9) < 0.01 * d["value"] + 1e-3


def test_bench_refuses_more_rccl_ranks_than_gpus():
    import torch
    n = torch.cuda.device_count()
    r = run_bench(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--bases", "1000000"], {"KMERS_BENCH_BACKEND": "nccl"})
    assert r.returncode != 0 and r.stdout.strip() == "" and "one GPU per rank" in r.stderr


def test_rank_started_with_the_wrong_world_size_fails():
    """A rank whose WORLD_SIZE differs from --gpus must not print a line for a job that is not the one asked for."""
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--bases", "1000000", "--no-other-configs", "--no-cpu-baseline", "--no-pmc"],
                  {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_async_launches_share_one_error_slot(km, ctx):
    """Several KMERS_ASYNC launches between two kmers_sync calls: the kernels record (position, symbol) at fault time,
    so the sync needs no sequence any more -- the smallest reported position wins, with the right symbol, even when the
    failing launch was not the last one and its source has been overwritten since."""
    import naive
    cap = km._capi
    rng = np.random.default_rng(5)
    K = 31
    clean = naive.random_text(rng, 50_000)
    bad = list(naive.random_text(rng, 3_000))
    bad[1234] = "N"
    bad[2500] = "R"
    w_clean = naive.longseq_words(clean, 4)
    w_bad = naive.longseq_words("".join(bad), 4)
    d_clean, d_bad = ctx.alloc(w_clean.nbytes + 8), ctx.alloc(w_bad.nbytes + 8)
    ctx.h2d(d_clean, w_clean)
    ctx.h2d(d_bad, w_bad)
    out = ctx.alloc(50_000 * 8)
    res = cap.Result()
    flags = cap.MEM_DEVICE | cap.ASYNC
    s_bad = cap.Seq(d_bad, len(bad), 0, 1_000_000, 4, 0)      # a shard whose positions start at 1 000 000
    s_clean = cap.Seq(d_clean, len(clean), 0, 0, 4, 0)
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(s_bad), K, 2, out, None, 0, flags, C.byref(res)) == 0
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(s_clean), K, 2, out, None, 0, flags, C.byref(res)) == 0
    ctx.h2d(d_bad, np.zeros_like(w_bad))                      # the failing source is gone before the sync
    rc, sres = ctx.sync()
    assert rc == cap.E_ENCODE and sres.err_pos == 1_000_000 + 1235 and sres.err_enc == 0xF, (rc, sres.err_pos, sres.err_enc)
    rc, sres = ctx.sync()                                      # the slot is re-armed
    assert rc == 0
    for p in (d_clean, d_bad, out):
        ctx.free(p)


def test_plain_c_sharded_client(tmp_path):
    """examples/sharded_canonical.c: the sharded path from plain C -- no torch in the process, so the RCCL the library was
    linked with (/opt/rocm) is the one that runs: shard plan, halo words through kmers_comm_sendrecv on a communicator,
    index_origin, the two reductions; 8 and 3 shards must reproduce the unsharded checksum."""
    csrc = os.path.join(ROOT, "kmers.jl_amd", "csrc")
    exe = tmp_path / "sharded_canonical"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "sharded_canonical.c"),
                    "-L", csrc, "-lkmers_hip", f"-Wl,-rpath,{csrc}", "-o", str(exe)], check=True)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    for shards, bases in (("8", "10000003"), ("3", "999"), ("5", "64")):
        out = subprocess.run([str(exe), shards, bases], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, (out.stdout, out.stderr)
        assert "communicator: rank 0 of 1" in out.stdout and ": equal" in out.stdout, out.stdout


def test_bench_rccl_code_path_on_one_rank():
    """Exactly the code a rank of `bench.py --gpus N` runs under RCCL -- process group on `nccl`, the ncclUniqueId handed over
    torch.distributed, kmers_comm_create, kmers_halo_exchange in every step, kmers_first_error_allreduce,
    kmers_offsets_allgather -- with the one rank a 1-GPU box admits (KMERS_BENCH_FORCE_GROUP=1 under torch.distributed.run)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, KMERS_BENCH_FORCE_GROUP="1", KMERS_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--bases", "8000000", "--no-other-configs", "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["verified"] is True
    assert d["config"]["backend"] == "nccl" and d["config"]["halo_transport"] == "native"
    assert "kmers_halo_exchange" in d["config"]["sharding"]


def test_bench_falls_back_to_torch_rccl_when_the_library_has_no_communicator():
    """`bench.py --gpus N` must not lose its measurement to the library's OWN communicator: if kmers_comm_create cannot be had
    (here: KMERS_RCCL_LIB=none, so the C ABI's communication answers KMERS_E_UNSUPPORTED), every rank agrees to move
    the halo with torch.distributed's all_gather over RCCL instead, and the line says so."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, KMERS_BENCH_FORCE_GROUP="1", KMERS_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0",
               KMERS_RCCL_LIB="none")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--bases", "8000000", "--no-other-configs", "--no-cpu-baseline", "--no-pmc"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["verified"] is True
    assert d["config"]["backend"] == "nccl" and d["config"]["halo_transport"] == "allgather"
    assert "kmers_comm_create failed" in r.stderr and "falls back" in r.stderr
