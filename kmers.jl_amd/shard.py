"""Contiguous sharding of one long LongSequence over the GPUs of a node, with the (K-1)-base
halo exchanged between neighbours over torch.distributed (backend "nccl" = RCCL over xGMI on
the GPU box, "gloo" in the CPU tests).

The reference has no distributed code; kmer i depends only on symbols
[i, i+K), so shard g owns a contiguous range of kmer START positions whose first symbol sits on a
source-word boundary, and needs the first ceil((K-1)*bits/64) words of shard g+1 appended to its
own words.  That one neighbour step is the only communication on the path; outputs stay on the
rank that produced them (concatenation in rank order == the reference's iteration order).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Shard:
    rank: int
    first_kmer: int      # global 0-based index of the first kmer start owned
    n_kmers: int         # kmers owned
    first_word: int      # global index of the first source word owned
    n_own_words: int     # source words owned (== words this rank generates / is handed)
    halo_words: int      # words received from rank+1 (0 on the last shard)
    n_bases: int         # symbols in this shard's view (own kmers' windows): n_kmers - 1 + K, 0 if empty
    send_words: int      # words sent to rank-1 (0 on the first shard)

    @property
    def first_base(self):
        return self.first_kmer  # stride 1: kmer i starts at symbol i


def plan_shards(n_bases, k, n_shards, src_bits):
    """Split kmer starts [0, n_bases-k+1) into n_shards contiguous word-aligned ranges."""
    if k < 1 or n_shards < 1 or src_bits not in (2, 4):
        raise ValueError("bad shard plan arguments")
    per_word = 64 // src_bits
    n_kmers = max(0, n_bases - k + 1)
    total_words = (n_bases * src_bits + 63) // 64
    # starts per shard: equal split rounded up to a whole number of source words
    per = -(-n_kmers // n_shards)
    per = -(-per // per_word) * per_word if per else per_word
    halo = ((k - 1) * src_bits + 63) // 64
    if n_shards > 1 and per // per_word < halo + 1:
        # too short to give every shard at least a halo's worth of words: shard 0 does it all
        first = Shard(0, 0, n_kmers, 0, total_words, 0, (n_kmers - 1 + k) if n_kmers else 0, 0)
        return [first] + [Shard(g, n_kmers, 0, total_words, 0, 0, 0, 0) for g in range(1, n_shards)]
    shards = []
    for g in range(n_shards):
        lo = min(n_kmers, g * per)
        hi = min(n_kmers, (g + 1) * per)
        fw = min(total_words, g * per // per_word)
        # own words: up to the next shard's first word (the last shard keeps the tail)
        lw = total_words if g == n_shards - 1 else min(total_words, (g + 1) * per // per_word)
        nk = hi - lo
        # words of the next shard this one needs to finish its last windows
        need_end = ((lo + nk - 1 + k) * src_bits + 63) // 64 if nk else fw
        h = max(0, min(halo, need_end - lw)) if g < n_shards - 1 else 0
        shards.append(Shard(g, lo, nk, fw, lw - fw, h, (nk - 1 + k) if nk else 0, 0))
    # what each rank sends = what its left neighbour needs
    out = []
    for g, s in enumerate(shards):
        send = shards[g - 1].halo_words if g > 0 else 0
        out.append(Shard(s.rank, s.first_kmer, s.n_kmers, s.first_word, s.n_own_words, s.halo_words,
                         s.n_bases, send))
    return out


def exchange_halo(buf, shard, group=None):
    """One neighbour step: append the first `halo_words` words of rank+1 to this rank's buffer.

    `buf` is a 1-D int64 tensor of n_own_words + halo_words elements whose first n_own_words hold
    this rank's words (CPU tensor under gloo, HBM tensor under nccl/RCCL).  Every rank of the
    group must call this.  Returns the list of posted requests already waited on."""
    import torch.distributed as dist
    ops = []
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if shard.send_words and rank > 0:
        ops.append(dist.P2POp(dist.isend, buf[:shard.send_words], rank - 1, group))
    if shard.halo_words and rank < world - 1:
        ops.append(dist.P2POp(dist.irecv, buf[shard.n_own_words:shard.n_own_words + shard.halo_words],
                              rank + 1, group))
    if not ops:
        return []
    reqs = dist.batch_isend_irecv(ops)
    for r in reqs:
        r.wait()
    return reqs
