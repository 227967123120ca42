"""Contiguous sharding of one long LongSequence over the GPUs of a node, with the (K-1)-base
halo exchanged between neighbours: on the GPU box through the C ABI's RCCL entry points
(`NativeComm`: kmers_halo_exchange / kmers_first_error_allreduce / kmers_offsets_allgather on the
context's stream, xGMI between the GPUs of a node; torch.distributed only carries the 128-byte
ncclUniqueId), and over torch.distributed itself ("gloo") in the CPU tests, which exercise the same
plan and the same semantics without a GPU (`HaloExchanger`, `first_error`, `output_offsets`).

The reference has no distributed code; kmer i depends only on symbols
[i, i+K), so shard g owns a contiguous range of kmer START positions whose first symbol sits on a
source-word boundary, and needs the first ceil((K-1)*bits/64) words of shard g+1 appended to its
own words.  That one neighbour step is the only communication on the path; outputs stay on the
rank that produced them (concatenation in rank order == the reference's iteration order).
"""
import math
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    first_kmer: int      # global 0-based ordinal of the first kmer owned
    n_kmers: int         # kmers owned
    first_word: int      # global index of the first source word owned
    n_own_words: int     # source words owned (== words this rank generates / is handed)
    halo_words: int      # words received from rank+1 (0 on the last shard)
    n_bases: int         # symbols in this shard's view: (n_kmers - 1) * stride + K, 0 if empty
    send_words: int      # words sent to rank-1 (0 on the first shard)
    stride: int = 1

    @property
    def first_base(self):
        return self.first_kmer * self.stride  # kmer i starts at symbol i * stride


def plan_shards(n_bases, k, n_shards, src_bits, stride=1):
    """Split the kmers of a length-n_bases sequence into n_shards contiguous ranges whose first symbol
    sits on a source-word boundary and on the stride lattice (the same arithmetic as the C ABI's
    kmers_shard_plan; tests hold the two equal)."""
    if k < 1 or n_shards < 1 or stride < 1 or src_bits not in (2, 4, 8):
        raise ValueError("bad shard plan arguments")
    per_word = 64 // src_bits
    n_kmers = (n_bases - k) // stride + 1 if n_bases >= k else 0
    total_words = (n_bases * src_bits + 63) // 64
    unit = per_word // math.gcd(stride, per_word)  # kmers per lcm(stride, per_word) symbols
    per = -(-n_kmers // n_shards)
    per = -(-per // unit) * unit if per else unit
    words_per_shard = per * stride // per_word
    halo = (max(0, k - stride) * src_bits + 63) // 64
    if n_kmers == 0 or (n_shards > 1 and words_per_shard < halo + 1):
        # too short to give every shard at least a halo's worth of words: shard 0 does it all
        first = Shard(0, 0, n_kmers, 0, total_words, 0, n_bases, 0, stride)
        return [first] + [Shard(g, n_kmers, 0, total_words, 0, 0, 0, 0, stride) for g in range(1, n_shards)]
    shards = []
    for g in range(n_shards):
        lo = min(n_kmers, g * per)
        hi = min(n_kmers, (g + 1) * per)
        fw = min(total_words, g * words_per_shard)
        # own words: up to the next shard's first word (the last shard keeps the tail)
        lw = total_words if g == n_shards - 1 else min(total_words, (g + 1) * words_per_shard)
        nk = hi - lo
        # words of the next shard this one needs to finish its last windows
        need_end = (((lo + nk - 1) * stride + k) * src_bits + 63) // 64 if nk else fw
        h = max(0, min(halo, need_end - lw)) if g < n_shards - 1 else 0
        shards.append(Shard(g, lo, nk, fw, lw - fw, h, ((nk - 1) * stride + k) if nk else 0, 0, stride))
    # what each rank sends = what its left neighbour needs
    out = []
    for g, s in enumerate(shards):
        send = shards[g - 1].halo_words if g > 0 else 0
        out.append(Shard(s.rank, s.first_kmer, s.n_kmers, s.first_word, s.n_own_words, s.halo_words,
                         s.n_bases, send, stride))
    return out


class HaloExchanger:
    """The one neighbour step of the path: after `exchange()`, words
    [n_own_words, n_own_words + halo_words) of this rank's buffer hold the first words of rank+1.

    `buf` is a 1-D int64 tensor (HBM tensor under nccl/RCCL, CPU tensor under gloo) whose first
    n_own_words elements are this rank's words.  Two transports, same result:
      "p2p"       ncclSend/ncclRecv between neighbours (batch_isend_irecv), the minimal exchange;
      "allgather" every rank contributes its first `width` words and keeps rank+1's piece -- the
                  default: one ordinary collective of a few dozen bytes, no per-pair communicator
                  set-up; the payload is latency-bound either way (<= 32 B per rank).
    Buffers are allocated once; exchange() allocates nothing."""

    def __init__(self, buf, shard, plan, group=None, transport=None):
        import os

        import torch
        import torch.distributed as dist
        self.dist, self.buf, self.shard, self.group = dist, buf, shard, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.transport = transport or os.environ.get("KMERS_HALO_TRANSPORT", "allgather")
        if self.transport not in ("allgather", "p2p"):
            raise ValueError(f"unknown halo transport {self.transport!r} (allgather, p2p; RCCL behind the C ABI: shard.NativeComm)")
        self.width = max([s.halo_words for s in plan] + [1])
        # gloo moves host memory: a buffer in HBM (ranks sharing one device while debugging) is staged through the host
        self.via_host = bool(self.world > 1 and buf.is_cuda and dist.get_backend(group) == "gloo")
        wdev = "cpu" if self.via_host else buf.device
        if self.world > 1:
            self.send = torch.zeros(self.width, dtype=buf.dtype, device=wdev)
            self.recv = torch.zeros(self.width, dtype=buf.dtype, device=wdev)
        if self.world > 1 and self.transport == "allgather":
            self.pieces = [torch.zeros(self.width, dtype=buf.dtype, device=wdev) for _ in range(self.world)]

    def exchange(self):
        if self.world == 1:
            return
        sh, buf, dist = self.shard, self.buf, self.dist
        if self.transport == "p2p":
            ops = []
            if sh.send_words and self.rank > 0:
                self.send[:sh.send_words].copy_(buf[:sh.send_words])
                ops.append(dist.P2POp(dist.isend, self.send[:sh.send_words], self.rank - 1, self.group))
            if sh.halo_words and self.rank < self.world - 1:
                ops.append(dist.P2POp(dist.irecv, self.recv[:sh.halo_words], self.rank + 1, self.group))
            if ops:
                for r in dist.batch_isend_irecv(ops):
                    r.wait()
            if sh.halo_words and self.rank < self.world - 1:
                buf[sh.n_own_words:sh.n_own_words + sh.halo_words].copy_(self.recv[:sh.halo_words])
            return
        n = min(self.width, sh.n_own_words)
        if n:
            self.send[:n].copy_(buf[:n])
        dist.all_gather(self.pieces, self.send, group=self.group)
        if sh.halo_words:
            buf[sh.n_own_words:sh.n_own_words + sh.halo_words].copy_(self.pieces[self.rank + 1][:sh.halo_words])


def exchange_halo(buf, shard, group=None, plan=None, transport=None):
    """One-shot form of HaloExchanger (allocates its workspace on every call)."""
    HaloExchanger(buf, shard, plan or [shard], group, transport).exchange()


_NO_ERROR = (1 << 63) - 1


def first_error(status, err_pos, err_enc, group=None, device="cpu"):
    """The reference throws at the FIRST ambiguous symbol in sequence order
    (FwKmers.jl:112, CanonicalKmers.jl:139): all_reduce(MIN) over the shards' (global 1-based
    position, raw encoding) pairs.  Returns (status, err_pos, err_enc) identical on every rank."""
    import torch
    import torch.distributed as dist
    key = ((int(err_pos) << 8) | (int(err_enc) & 0xFF)) if status == 1 else _NO_ERROR
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.tensor([key], dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        key = int(t.item())
    if key == _NO_ERROR:
        return status if status != 1 else 0, 0, 0
    return 1, key >> 8, key & 0xFF


def output_offsets(n_local, group=None, device="cpu"):
    """Exclusive scan of the shards' element counts (UnambiguousKmers has SizeUnknown,
    UnambiguousKmers.jl:33): returns (offset of this rank's elements in the global output, total)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0, int(n_local)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([int(n_local)], dtype=torch.int64, device=device), group=group)
    counts = [int(c.item()) for c in counts]
    return sum(counts[:rank]), sum(counts)


class NativeComm:
    """The RCCL communicator of the C ABI (include/kmers_hip.h, "the communication of the sharded path").
    rank r of the communicator runs shard r.  `bootstrap(ctx)` makes one over the ranks of an initialised
    torch.distributed group of ANY backend: rank 0 asks RCCL for an ncclUniqueId and the 128 bytes travel
    over the group's store -- torch moves no sequence data."""

    def __init__(self, ctx, handle, rank, n_ranks):
        self.ctx, self.handle, self.rank, self.n_ranks = ctx, handle, rank, n_ranks

    @classmethod
    def create(cls, ctx, id_bytes, n_ranks, rank):
        import ctypes as C

        from . import _capi as cap
        if len(id_bytes) != cap.COMM_ID_BYTES:
            raise ValueError("an ncclUniqueId has 128 bytes")
        h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), cap.COMM_ID_BYTES)
        ctx.check(ctx.lib.kmers_comm_create(ctx.handle, buf, n_ranks, rank, C.byref(h)), "kmers_comm_create")
        return cls(ctx, h, rank, n_ranks)

    @staticmethod
    def new_id(lib):
        import ctypes as C

        from . import _capi as cap
        buf = C.create_string_buffer(cap.COMM_ID_BYTES)
        rc = lib.kmers_comm_id(buf)
        if rc != 0:
            raise RuntimeError(f"kmers_comm_id failed with {cap.STATUS_NAMES.get(rc, rc)}")
        return bytes(buf.raw)

    @classmethod
    def bootstrap(cls, ctx, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls.create(ctx, cls.new_id(ctx.lib), 1, 0)
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [cls.new_id(ctx.lib) if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        return cls.create(ctx, box[0], world, rank)

    def _shard_struct(self, shard):
        from . import _capi as cap
        return cap.ShardPlan(shard.first_kmer, shard.n_kmers, shard.first_base, shard.n_bases, shard.first_word,
                             shard.n_own_words, shard.halo_words, shard.send_words)

    def halo_exchange(self, shard, words_ptr):
        """Enqueue the neighbour step on the context's stream (kmers_halo_exchange)."""
        import ctypes as C
        cs = shard if not isinstance(shard, Shard) else self._shard_struct(shard)
        self.ctx.check(self.ctx.lib.kmers_halo_exchange(self.ctx.handle, self.handle, C.byref(cs), words_ptr), "kmers_halo_exchange")

    def sendrecv(self, send_ptr, send_words, send_peer, recv_ptr, recv_words, recv_peer):
        self.ctx.check(self.ctx.lib.kmers_comm_sendrecv(self.ctx.handle, self.handle, send_ptr, send_words, send_peer,
                                                        recv_ptr, recv_words, recv_peer), "kmers_comm_sendrecv")

    def first_error(self, status, err_pos=0, err_enc=0):
        """kmers_first_error_allreduce: (status, err_pos, err_enc) of the first EncodeError over all shards."""
        import ctypes as C

        from . import _capi as cap
        res = cap.Result(int(status), int(err_enc), int(err_pos), 0)
        rc = self.ctx.lib.kmers_first_error_allreduce(self.ctx.handle, self.handle, C.byref(res))
        if rc not in (cap.OK, cap.E_ENCODE):
            self.ctx.check(rc, "kmers_first_error_allreduce")
        return int(res.status), int(res.err_pos), int(res.err_enc)

    def output_offsets(self, n_local):
        """kmers_offsets_allgather: (offset of this shard's elements in the global output, total)."""
        import ctypes as C
        off, tot = C.c_uint64(), C.c_uint64()
        self.ctx.check(self.ctx.lib.kmers_offsets_allgather(self.ctx.handle, self.handle, int(n_local), C.byref(off), C.byref(tot)),
                       "kmers_offsets_allgather")
        return int(off.value), int(tot.value)

    def close(self):
        if self.handle:
            self.ctx.lib.kmers_comm_destroy(self.ctx.handle, self.handle)
            self.handle = None
