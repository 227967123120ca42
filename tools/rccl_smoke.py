#!/usr/bin/env python3
"""RCCL API smoke on one GPU (world_size 1): the exact calls the multi-rank bench makes
(init_process_group("nccl", device_id), all_gather / P2P-free barrier on the library's stream)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
import kmers_jl_amd as km
from kmers_jl_amd.shard import HaloExchanger, plan_shards

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ctx = km.Context(0)
stream = torch.cuda.ExternalStream(ctx.lib.kmers_ctx_stream(ctx.handle), device=dev)
plan = plan_shards(10_000_000, 31, 1, 4)
with torch.cuda.stream(stream):
    buf = torch.arange(plan[0].n_own_words + 4, dtype=torch.int64, device=dev)
    hx = HaloExchanger(buf, plan[0], plan)
    hx.world = 1
    # force the collective path once even though a single rank has no neighbour
    send = torch.arange(2, dtype=torch.int64, device=dev)
    pieces = [torch.zeros(2, dtype=torch.int64, device=dev)]
    dist.all_gather(pieces, send)
    t = torch.ones(1, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize()
dist.barrier()
assert pieces[0].tolist() == [0, 1]
print("rccl smoke ok:", torch.cuda.get_device_name(0), "nccl", torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.destroy_process_group()
