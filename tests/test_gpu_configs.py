"""BASELINE.json configs at their FULL sizes on one MI355X, through the C ABI with resident data.
Outputs this large cannot be compared element by element with the CPU oracle in reasonable time,
so each config is checked by size-independent properties evaluated on the GPU over EVERY element
(hash/canonical/rolling/involution identities, a checksum of checksums across shards) plus oracle
comparisons of windows at the places that matter (start, end, shard and tile boundaries)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

FX = 0x517CC1B727220A95
GOLDEN = 0x9E3779B97F4A7C15


@pytest.fixture(scope="module")
def km():
    import kmers_jl_amd
    return kmers_jl_amd


@pytest.fixture(scope="module")
def ctx(km):
    c = km.Context(0)
    yield c
    c.close()


def dev_empty(n):
    return torch.empty(int(n), dtype=torch.int64, device="cuda:0")


def synth(ctx, seed, first_word, n_words, bits, amb=0):
    buf = dev_empty(n_words + 2)
    ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, first_word, n_words, bits, amb, buf.data_ptr()), "synth")
    return buf


def xor_fold(t):
    while t.numel() > 1:
        h = t.numel() // 2
        rest = t[2 * h:]
        t = torch.bitwise_xor(t[:h], t[h:2 * h])
        if rest.numel():
            t = torch.cat([t, rest])
    return int(t.item()) & (2**64 - 1)


def chunks(n, step=1 << 27):
    for lo in range(0, n, step):
        yield lo, min(n, lo + step)


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


def test_c1_fw21_one_mbase_full_compare(km, ctx, orc):
    """configs[0]: FwDNAMers{21} over 1 Mbase LongDNA{4} -- small enough for a full comparison."""
    cap = km._capi
    L, K, bits = 1_000_000, 21, 4
    seed = GOLDEN ^ 1
    nw = (L * bits + 63) // 64
    buf = synth(ctx, seed, 0, nw, bits)
    out = dev_empty(L - K + 1)
    res = cap.Result()
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, out.data_ptr(), None, cap.MEM_DEVICE, C.byref(res)) == 0
    assert res.n_out == 999_980
    ek, _ = orc.fw_kmers(orc.synth_words(seed, 0, nw, bits), L, bits, 2, K)
    assert np.array_equal(host_u64(out), ek[:, 0])


def test_c2_canonical31_hash_one_gbase(km, ctx, orc):
    """configs[1]: CanonicalDNAMers{31} + fx_hash over 1 Gbase LongDNA{4}."""
    cap = km._capi
    L, K, bits = 1_000_000_000, 31, 4
    seed = GOLDEN ^ 2
    n = L - K + 1
    nw = (L * bits + 63) // 64
    buf = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    ck, hs, fw, rc = dev_empty(n), dev_empty(n), dev_empty(n), dev_empty(n)
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, ck.data_ptr(), hs.data_ptr(), 0, cap.MEM_DEVICE,
                                   C.byref(res)) == 0 and res.n_out == n
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, fw.data_ptr(), rc.data_ptr(), cap.MEM_DEVICE,
                            C.byref(res)) == 0
    mask = (1 << 62) - 1
    cmul = torch.tensor(FX, dtype=torch.int64, device="cuda:0")
    folded = 0
    for lo, hi in chunks(n):
        f, r, c, h = fw[lo:hi], rc[lo:hi], ck[lo:hi], hs[lo:hi]
        assert bool(torch.equal(h, c * cmul))                              # fx_hash(x) = x * C for one word, seed 0
        assert bool(torch.equal(c, torch.minimum(f, r)))                   # fw < rv ? fw : rv (62-bit values: signed == unsigned)
        assert int(torch.max(f)) <= mask and int(torch.max(r)) <= mask     # unused top bits are zero
        hi2 = min(n, hi + 1)
        fa, fb = fw[lo:hi2 - 1], fw[lo + 1:hi2]
        assert bool(torch.equal(fb >> 2, fa & (mask >> 2)))                # shift_encoding: window moves by one symbol
        ra, rb = rc[lo:hi2 - 1], rc[lo + 1:hi2]
        assert bool(torch.equal(rb & (mask >> 2), ra >> 2))                # shift_first_encoding on the other strand
        assert bool(torch.equal((rb >> 60) ^ 3, fb & 3))                   # the symbol entering rc is the complement
        folded ^= xor_fold(c.contiguous())
    xr = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)) == 0
    assert folded == xr.value
    # canonical is idempotent and agrees with the element-wise transform
    again = dev_empty(n)
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_CANONICAL, fw.data_ptr(), K, 2, n, again.data_ptr(), cap.MEM_DEVICE) == 0
    assert bool(torch.equal(again, ck))
    # oracle windows: start, an unaligned middle, the end
    per = 64 // bits
    probe = 1 << 20
    for first in (0, (n // 3 // per) * per, ((n - probe) // per) * per):
        nb = min(probe, n - first) + K - 1
        w = orc.synth_words(seed, first // per, (nb * bits + 63) // 64 + 1, bits)
        ek, eh, _ = orc.canonical(w, nb, bits, 2, K)
        assert np.array_equal(host_u64(ck[first:first + len(ek)]), ek[:, 0])
        assert np.array_equal(host_u64(hs[first:first + len(eh)]), eh)


def test_north_star_canonical31_hash_ten_gbase_one_gpu(km, ctx, orc):
    """BASELINE.json north_star: canonical 31-mers + fx_hash over 10 Gbase LongDNA{4} on ONE GPU (5 GB in, 160 GB of kmers
    and hashes out; CanonicalKmers.jl:131-144, kmer.jl:255-261).  Properties over all 9 999 999 970 elements, the fused
    reducer as a checksum, oracle windows at the start, in the middle, across the 2^32-element boundary and at the end."""
    cap = km._capi
    L, K, bits = 10_000_000_000, 31, 4
    torch.cuda.empty_cache()
    free_b, total_b = torch.cuda.mem_get_info(0)
    if free_b < 172e9:
        # an MI355X has 288 GB: if 165 GB are not free there, something (an earlier test, another process) is holding memory
        # it should not, and a skip would read as green
        assert total_b < 256e9, f"the north-star test needs 165 GB of HBM and only {free_b / 1e9:.0f} of {total_b / 1e9:.0f} GB are free"
        pytest.skip(f"needs 165 GB of HBM, this device has {total_b / 1e9:.0f} GB")
    seed = GOLDEN ^ 10
    n = L - K + 1
    nw = (L * bits + 63) // 64
    buf = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    ck, hs = dev_empty(n), dev_empty(n)
    assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, ck.data_ptr(), hs.data_ptr(), 0, cap.MEM_DEVICE,
                                   C.byref(res)) == 0 and res.n_out == n
    mask = (1 << 62) - 1
    cmul = torch.tensor(FX, dtype=torch.int64, device="cuda:0")
    folded = 0
    for lo, hi in chunks(n, 1 << 28):
        c, h = ck[lo:hi], hs[lo:hi]
        assert bool(torch.equal(h, c * cmul))                              # fx_hash(x) = x * C for one word, seed 0
        assert int(torch.max(c)) <= mask and int(torch.min(c)) >= 0        # unused top bits are zero
        folded ^= xor_fold(c.contiguous())
    xr = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)) == 0
    assert folded == xr.value
    # canonical kmers are fixed points of canonical(): idempotence over the first 2^30 elements via the element-wise transform
    m = 1 << 30
    again = dev_empty(m)
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_CANONICAL, ck.data_ptr(), K, 2, m, again.data_ptr(), cap.MEM_DEVICE) == 0
    assert bool(torch.equal(again, ck[:m]))
    del again
    per = 64 // bits
    probe = 1 << 20
    for first in (0, (n // 3 // per) * per, (1 << 32) - (probe // 2), ((n - probe) // per) * per):
        first = first // per * per
        nb = min(probe, n - first) + K - 1
        w = orc.synth_words(seed, first // per, (nb * bits + 63) // 64 + 1, bits)
        ek, eh, _ = orc.canonical(w, nb, bits, 2, K)
        assert np.array_equal(host_u64(ck[first:first + len(ek)]), ek[:, 0]), first
        assert np.array_equal(host_u64(hs[first:first + len(eh)]), eh), first


def test_c3_canonical31_ten_gbase_two_bit_eight_shards(km, ctx, orc):
    """configs[2]: CanonicalDNAMers{31} over 10 Gbase LongDNA{2} sharded 8 ways with a (K-1)-base halo.
    The 8 shards run one after another on this device exactly as 8 ranks would (own words + the halo
    words of the next shard); the XOR of the shard checksums must equal the checksum of one pass over
    the whole sequence, and the kmers that straddle every shard boundary are compared with the oracle."""
    from kmers_jl_amd.shard import plan_shards
    cap = km._capi
    L, K, bits, world = 10_000_000_000, 31, 2, 8
    seed = GOLDEN ^ 3
    plan = plan_shards(L, K, world, bits)
    assert sum(s.n_kmers for s in plan) == L - K + 1
    res = cap.Result()
    total_xor = 0
    out = dev_empty(max(s.n_kmers for s in plan))
    for sh in plan:
        buf = dev_empty(sh.n_own_words + sh.halo_words + 2)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words, bits, 0, buf.data_ptr()), "synth")
        if sh.halo_words:  # what rank+1 would send: its first halo_words words
            ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word + sh.n_own_words, sh.halo_words, bits, 0,
                                              buf.data_ptr() + 8 * sh.n_own_words), "synth halo")
        seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, bits, 0)
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, out.data_ptr(), None, 0, cap.MEM_DEVICE,
                                       C.byref(res)) == 0 and res.n_out == sh.n_kmers
        for lo, hi in chunks(sh.n_kmers):
            total_xor ^= xor_fold(out[lo:hi].contiguous())
        # the last 4096 kmers of the shard use the halo: compare with the oracle on the global sequence
        tail = 4096
        first = sh.first_kmer + sh.n_kmers - tail
        fw_word = first // 32
        nb = tail + K - 1 + (first - fw_word * 32)
        w = orc.synth_words(seed, fw_word, (nb * bits + 63) // 64 + 1, bits)
        ek, _, _ = orc.canonical(w, nb, bits, 2, K, hashes=False)
        assert np.array_equal(host_u64(out[sh.n_kmers - tail:sh.n_kmers]), ek[first - fw_word * 32:, 0][:tail])
        del buf
    # one pass over the whole 10 Gbase (2.5 GB of words) with the fused consumer
    nw = (L * bits + 63) // 64
    whole = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(whole.data_ptr(), L, 0, 0, bits, 0)
    xr = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)) == 0
    assert total_xor == xr.value


def test_north_star_ten_gbase_four_bit_eight_shards(km, ctx, orc):
    """The north star's own split: CanonicalDNAMers{31} + fx_hash over ONE 10 Gbase LongDNA{4} sequence as the 8 shards
    `bench.py --gpus 8 --total-bases 10000000000` gives its ranks (kmers_shard_plan: contiguous kmer ranges on word
    boundaries, 2 halo words from the next shard).  The shards run one after another on this device exactly as 8 ranks would;
    the XOR of the shard checksums must equal one fused pass over the whole sequence, hashes == kmers * FX over every element
    of every shard, and the kmers AND hashes that straddle every shard boundary are compared with the oracle on the global
    sequence (CanonicalKmers.jl:131-144, kmer.jl:255-261)."""
    from kmers_jl_amd.shard import plan_shards
    cap = km._capi
    L, K, bits, world = 10_000_000_000, 31, 4, 8
    per = 64 // bits
    seed = GOLDEN ^ 10
    plan = plan_shards(L, K, world, bits)
    assert sum(s.n_kmers for s in plan) == L - K + 1 and all(s.halo_words == 2 for s in plan[:-1]) and plan[-1].halo_words == 0
    # the C ABI's plan is the same plan
    for g, s in enumerate(plan):
        c = cap.ShardPlan()
        assert ctx.lib.kmers_shard_plan(L, K, 1, bits, world, g, C.byref(c)) == 0
        assert (c.first_kmer, c.n_kmers, c.first_word, c.n_own_words, c.halo_words, c.send_words) == \
            (s.first_kmer, s.n_kmers, s.first_word, s.n_own_words, s.halo_words, s.send_words)
    res = cap.Result()
    total_xor = 0
    cmul = torch.tensor(FX, dtype=torch.int64, device="cuda:0")
    nmax = max(s.n_kmers for s in plan)
    out, hs = dev_empty(nmax), dev_empty(nmax)
    for sh in plan:
        buf = dev_empty(sh.n_own_words + sh.halo_words + 2)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words, bits, 0, buf.data_ptr()), "synth")
        if sh.halo_words:  # what rank+1 sends: its first halo_words words
            ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word + sh.n_own_words, sh.halo_words, bits, 0,
                                              buf.data_ptr() + 8 * sh.n_own_words), "synth halo")
        seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_kmer, bits, 0)
        assert ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, out.data_ptr(), hs.data_ptr(), 0, cap.MEM_DEVICE,
                                       C.byref(res)) == 0 and res.n_out == sh.n_kmers
        for lo, hi in chunks(sh.n_kmers):
            assert bool(torch.equal(hs[lo:hi], out[lo:hi] * cmul))
            total_xor ^= xor_fold(out[lo:hi].contiguous())
        # both ends of the shard against the oracle on the GLOBAL sequence: the first 4096 kmers (the previous shard's halo
        # ends inside them) and the last 4096 (which read this shard's halo)
        tail = 4096
        for first in (sh.first_kmer, sh.first_kmer + sh.n_kmers - tail):
            fw_word = first // per
            skip = first - fw_word * per
            nb = tail + K - 1 + skip
            w = orc.synth_words(seed, fw_word, (nb * bits + 63) // 64 + 1, bits)
            ek, eh, _ = orc.canonical(w, nb, bits, 2, K)
            lo = first - sh.first_kmer
            assert np.array_equal(host_u64(out[lo:lo + tail]), ek[skip:, 0][:tail]), (sh.rank, first)
            assert np.array_equal(host_u64(hs[lo:lo + tail]), eh[skip:][:tail]), (sh.rank, first)
        del buf
    del out, hs
    torch.cuda.empty_cache()
    # one fused pass over the whole 10 Gbase (5 GB of words)
    nw = (L * bits + 63) // 64
    whole = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(whole.data_ptr(), L, 0, 0, bits, 0)
    xr = C.c_uint64()
    assert ctx.lib.kmers_reduce_xor(ctx.handle, C.byref(seq), K, 2, 1, C.byref(xr), cap.MEM_DEVICE, C.byref(res)) == 0
    assert total_xor == xr.value


def test_c4_fw63_revcomp_one_gbase(km, ctx, orc):
    """configs[3]: FwDNAMers{63} (two-word kmers) + reverse_complement over 1 Gbase LongDNA{4}."""
    cap = km._capi
    L, K, bits = 1_000_000_000, 63, 4
    seed = GOLDEN ^ 4
    n = L - K + 1
    nw = (L * bits + 63) // 64
    buf = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    fw, rc, tmp = dev_empty(2 * n), dev_empty(2 * n), dev_empty(2 * n)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, fw.data_ptr(), rc.data_ptr(), cap.MEM_DEVICE,
                            C.byref(res)) == 0 and res.n_out == n
    # reverse_complement(fw) == rc and reverse_complement is an involution (transformations.jl:32-34)
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, fw.data_ptr(), K, 2, n, tmp.data_ptr(), cap.MEM_DEVICE) == 0
    assert bool(torch.equal(tmp, rc))
    assert ctx.lib.kmers_transform(ctx.handle, cap.OP_REVCOMP, rc.data_ptr(), K, 2, n, tmp.data_ptr(), cap.MEM_DEVICE) == 0
    assert bool(torch.equal(tmp, fw))
    # head words use 62 bits; rolling relation across the two words of consecutive kmers
    f = fw.view(-1, 2)
    M62 = (1 << 62) - 1
    for lo, hi in chunks(n - 1):
        a, b = f[lo:hi], f[lo + 1:hi + 1]
        assert int(torch.max(a[:, 0])) <= M62 and int(torch.min(a[:, 0])) >= 0
        # shift_encoding on a two-word kmer (leftshift_carry, tuple_bitflipping.jl:24-33):
        # lo' = lo << 2 | code ; hi' = (hi << 2 | lo >> 62) & mask
        assert bool(torch.equal((b[:, 1] >> 2) & M62, a[:, 1] & M62))
        assert bool(torch.equal(b[:, 0], ((a[:, 0] << 2) | ((a[:, 1] >> 62) & 3)) & M62))
    per = 64 // bits
    probe = 1 << 19
    for first in (0, ((n - probe) // per) * per):
        nb = min(probe, n - first) + K - 1
        w = orc.synth_words(seed, first // per, (nb * bits + 63) // 64 + 1, bits)
        efw, erv, _ = orc.fwrv(w, nb, bits, 2, K)
        assert np.array_equal(host_u64(f[first:first + len(efw)]), efw)
        assert np.array_equal(host_u64(rc.view(-1, 2)[first:first + len(erv)]), erv)


def test_c5_spaced21_3_one_gbase_strict_and_skip(km, ctx, orc):
    """configs[4]: SpacedDNAMers{21,3} over 1 Gbase LongDNA{4}.  Strict (the reference's SpacedKmers)
    on a clean sequence and on one with N at p = 0.04 (must raise at the first N inside the lattice),
    and the labelled skip variant = UnambiguousDNAMers{21} restricted to the stride lattice."""
    cap = km._capi
    L, K, J, bits = 1_000_000_000, 21, 3, 4
    seed = GOLDEN ^ 5
    nw = (L * bits + 63) // 64
    n = (L - K) // J + 1
    clean = synth(ctx, seed, 0, nw, bits)
    seq = cap.Seq(clean.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    sp = dev_empty(n)
    assert ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, sp.data_ptr(), cap.MEM_DEVICE, C.byref(res)) == 0
    assert res.n_out == n == 333_333_327
    # every third forward kmer
    fw = dev_empty(L - K + 1)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq), K, 2, fw.data_ptr(), None, cap.MEM_DEVICE, C.byref(res)) == 0
    assert bool(torch.equal(sp, fw[::J]))
    del fw
    # ambiguous copy: strict semantics throw at the first N (every symbol up to the last kmer is inspected)
    amb = synth(ctx, seed, 0, nw, bits, 2621)
    seq_a = cap.Seq(amb.data_ptr(), L, 0, 0, bits, 0)
    tmp = dev_empty(n)
    rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq_a), K, J, 2, tmp.data_ptr(), cap.MEM_DEVICE, C.byref(res))
    w = orc.synth_words(seed, 0, 4096, bits, 2621)
    _, eres = orc.spaced(w, 4096 * 16, bits, 2, K, J)
    assert rc == cap.E_ENCODE and eres.status == 1
    assert (res.err_pos, res.err_enc) == (eres.err_pos, eres.err_enc)
    # skip variant: windows on the lattice without ambiguous symbols, with their 1-based starts
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq_a), K, J, None, None, 0, cap.MEM_DEVICE, C.byref(res)) == 0
    m = int(res.n_out)
    kmers, starts = dev_empty(m), dev_empty(m)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq_a), K, J, kmers.data_ptr(), starts.data_ptr(), m,
                                     cap.MEM_DEVICE, C.byref(res)) == 0 and res.n_out == m
    assert 0.40 * n < m < 0.45 * n                                    # (1 - 0.04)^21 = 0.424 of the lattice
    assert bool(torch.all(starts[1:] > starts[:-1]))                  # iteration order
    assert bool(torch.all((starts - 1) % J == 0))
    assert bool(torch.equal(kmers, sp[(starts - 1) // J]))            # same kmer as strict Spaced on the clean copy
    # exact comparison with the oracle on the first 8 Mbase
    Lp = 8_000_000
    w = orc.synth_words(seed, 0, Lp * bits // 64 + 1, bits, 2621)
    ek, es, _ = orc.unambiguous(w, Lp, bits, K)
    keep = (es - 1) % J == 0
    cnt = int(keep.sum())
    assert np.array_equal(host_u64(kmers[:cnt]), ek[keep][:, 0])
    assert np.array_equal(starts[:cnt].cpu().numpy(), es[keep])


def test_unambiguous31_one_gbase_every_element(km, ctx, orc):
    """UnambiguousDNAMers{31} over 1 Gbase LongDNA{4} with N at p = 0.04 (the reference's own skipping iterator at the headline
    K and size), device outputs, checked over EVERY element without the oracle: the starts are exactly the positions whose
    31 symbols hold no N (N flags decoded from the source words and window-summed with torch), in order, and every kmer is
    the forward kmer of the clean copy at that start.  The oracle then confirms the first 4 Mbase bit for bit, and the tuple
    layout gives the same elements."""
    cap = km._capi
    L, K, bits = 1_000_000_000, 31, 4
    seed = GOLDEN ^ 6
    nw = (L * bits + 63) // 64
    amb = synth(ctx, seed, 0, nw, bits, 2621)
    seq_a = cap.Seq(amb.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq_a), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)) == 0
    m = int(res.n_out)
    assert 0.27 * L < m < 0.30 * L                                    # (1 - 0.04)^31 = 0.282
    kmers, starts = dev_empty(m), dev_empty(m)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq_a), K, 1, kmers.data_ptr(), starts.data_ptr(), m,
                                     cap.MEM_DEVICE, C.byref(res)) == 0 and res.n_out == m
    # the expected starts, from the source words alone: symbol j of word w is N iff its nibble is 0b1111
    n_cand = L - K + 1
    expected = []
    carry = torch.zeros(K - 1, dtype=torch.int32, device="cuda:0")      # N flags of the K-1 symbols before the chunk
    step = 1 << 26                                                      # symbols per chunk (a multiple of 16)
    found = 0
    for lo in range(0, L, step):
        hi = min(L, lo + step)
        words = amb[lo // 16:(hi + 15) // 16]
        nib = (words.unsqueeze(1) >> (4 * torch.arange(16, device="cuda:0", dtype=torch.int64))) & 0xF
        is_n = (nib == 0xF).reshape(-1)[:hi - lo].to(torch.int32)
        flags = torch.cat([carry, is_n])                                # symbols lo-(K-1) .. hi-1
        cs = torch.cumsum(flags, 0, dtype=torch.int32)
        cs = torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda:0"), cs])
        win = cs[K:] - cs[:-K]                                          # window sums of the starts lo-(K-1) .. hi-K
        first_start = lo - (K - 1)                                      # 0-based start of win[0]
        ok = torch.nonzero(win == 0).reshape(-1) + first_start
        ok = ok[(ok >= 0) & (ok < n_cand)]
        got = starts[found:found + ok.numel()]
        assert got.numel() == ok.numel() and bool(torch.equal(got, ok + 1)), lo
        found += ok.numel()
        carry = flags[-(K - 1):].clone()
        del nib, is_n, flags, cs, win, ok
    assert found == m
    # every kmer = the forward kmer of the clean copy (same seed, no N) at its start
    clean = synth(ctx, seed, 0, nw, bits)
    seq_c = cap.Seq(clean.data_ptr(), L, 0, 0, bits, 0)
    fw = dev_empty(n_cand)
    assert ctx.lib.kmers_fw(ctx.handle, C.byref(seq_c), K, 2, fw.data_ptr(), None, cap.MEM_DEVICE, C.byref(res)) == 0
    for lo, hi in chunks(m):
        assert bool(torch.equal(kmers[lo:hi], fw[starts[lo:hi] - 1])), lo
    del fw, clean
    # Tuple{Kmer,Int} elements: the same pairs interleaved
    tup = dev_empty(2 * m)
    assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq_a), K, 1, tup.data_ptr(), None, m,
                                     cap.MEM_DEVICE | cap.OUT_TUPLES, C.byref(res)) == 0 and res.n_out == m
    assert bool(torch.equal(tup[0::2], kmers)) and bool(torch.equal(tup[1::2], starts))
    # and the oracle on the first 4 Mbase
    Lp = 4_000_000
    w = orc.synth_words(seed, 0, Lp * bits // 64 + 1, bits, 2621)
    ek, es, _ = orc.unambiguous(w, Lp, bits, K)
    assert np.array_equal(host_u64(kmers[:len(ek)]), ek[:, 0]) and np.array_equal(starts[:len(es)].cpu().numpy(), es)


@pytest.mark.parametrize("world", [3, 8])
def test_spaced_and_unambiguous_over_logical_shards(km, ctx, orc, world):
    """SURVEY.md 8(e) for the two iterators whose shards are not plain kmer ranges: SpacedDNAMers{21,3} (shard boundaries on
    the stride lattice AND on source words) and UnambiguousDNAMers{21} (per-shard compaction + exclusive scan of the counts;
    global 1-based starts through index_origin).  The shards run one after another on this device exactly as `world` ranks
    would (own words + the halo words of the next shard, index_origin = first_base); their concatenation must equal one call
    over the whole sequence, and the strict iterator's first EncodeError is the minimum over the shards."""
    from kmers_jl_amd.shard import plan_shards
    cap = km._capi
    L, K, J, bits = 48_000_017, 21, 3, 4
    seed = GOLDEN ^ 8
    nw = (L * bits + 63) // 64
    res = cap.Result()

    def shard_buffer(sh, amb):
        buf = dev_empty(sh.n_own_words + sh.halo_words + 2)
        ctx.check(ctx.lib.kmers_synth_dna(ctx.handle, seed, sh.first_word, sh.n_own_words + sh.halo_words, bits, amb, buf.data_ptr()), "synth")
        return buf

    for amb in (0, 2621):
        whole = synth(ctx, seed, 0, nw, bits, amb)
        wseq = cap.Seq(whole.data_ptr(), L, 0, 0, bits, 0)
        # ---- SpacedDNAMers{21,3}
        n = (L - K) // J + 1
        ref = dev_empty(n)
        rc_whole = ctx.lib.kmers_spaced(ctx.handle, C.byref(wseq), K, J, 2, ref.data_ptr(), cap.MEM_DEVICE, C.byref(res))
        whole_err = (res.err_pos, res.err_enc)
        plan = plan_shards(L, K, world, bits, J)
        assert sum(sh.n_kmers for sh in plan) == n
        parts, errs = [], []
        for sh in plan:
            assert sh.first_base % J == 0 and (sh.first_base * bits) % 64 == 0
            buf = shard_buffer(sh, amb)
            out = dev_empty(max(sh.n_kmers, 1))
            seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_base, bits, 0)
            rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, J, 2, out.data_ptr(), cap.MEM_DEVICE, C.byref(res))
            if rc == 0:
                assert res.n_out == sh.n_kmers
                parts.append(out[:sh.n_kmers])
            else:
                assert rc == cap.E_ENCODE
                errs.append((res.err_pos, res.err_enc))
        if amb == 0:
            assert rc_whole == 0 and not errs and bool(torch.equal(torch.cat(parts), ref))
        else:
            assert rc_whole == cap.E_ENCODE and min(errs) == whole_err   # the reduction kmers_first_error_allreduce does
        # ---- UnambiguousDNAMers{21}: counts scanned over the shards, starts global
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(wseq), K, 1, None, None, 0, cap.MEM_DEVICE, C.byref(res)) == 0
        m = int(res.n_out)
        rk, rs = dev_empty(m), dev_empty(m)
        assert ctx.lib.kmers_unambiguous(ctx.handle, C.byref(wseq), K, 1, rk.data_ptr(), rs.data_ptr(), m, cap.MEM_DEVICE, C.byref(res)) == 0
        plan = plan_shards(L, K, world, bits, 1)
        gk, gs = dev_empty(m), dev_empty(m)
        offset = 0
        for sh in plan:
            buf = shard_buffer(sh, amb)
            seq = cap.Seq(buf.data_ptr(), sh.n_bases, 0, sh.first_base, bits, 0)
            # the shard writes straight into its slice of the global arrays (offset = exclusive scan of the counts)
            rc = ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, 1, gk.data_ptr() + 8 * offset, gs.data_ptr() + 8 * offset,
                                           m - offset, cap.MEM_DEVICE, C.byref(res))
            assert rc == 0
            offset += int(res.n_out)
        assert offset == m and bool(torch.equal(gk, rk)) and bool(torch.equal(gs, rs))
    # and the oracle on the head of the whole-sequence result (amb = 2621 is the last state of rk / rs)
    Lp = 2_000_000
    w = orc.synth_words(seed, 0, Lp * bits // 64 + 1, bits, 2621)
    ek, es, _ = orc.unambiguous(w, Lp, bits, K)
    assert np.array_equal(host_u64(rk[:len(ek)]), ek[:, 0]) and np.array_equal(rs[:len(es)].cpu().numpy(), es)


def test_synth_10k_fixture_on_device(km, ctx):
    """tests/golden/synth_10k.json: device generator + HIP iterators against the committed values
    (no oracle at run time)."""
    import ctypes as C
    import json
    import os
    cap = km._capi
    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "synth_10k.json")))
    L, seed = fx["n_bases"], int(fx["seed"], 16)
    for c in fx["cases"]:
        bits, K = c["src_bits"], c["K"]
        res = cap.Result()
        if c["iter"] == "canonical":
            nw = (L * bits + 63) // 64
            buf = synth(ctx, seed, 0, nw, bits)
            src = host_u64(buf)
            assert [f"0x{int(x):016x}" for x in src[:4]] == c["source_words_first4"]
            assert int(np.bitwise_xor.reduce(src[:nw])) == int(c["source_xor"], 16)
            N = (2 * K + 63) // 64
            n = L - K + 1
            km_d, hs_d = dev_empty(n * N), dev_empty(n)
            seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
            ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, km_d.data_ptr(), hs_d.data_ptr(), 0,
                                              cap.MEM_DEVICE, C.byref(res)), "canonical")
            kmers, hs = host_u64(km_d).reshape(n, N), host_u64(hs_d)
            assert res.n_out == c["n"] == n
            assert [int(np.bitwise_xor.reduce(kmers[:, j])) for j in range(N)] == [int(x, 16) for x in c["kmer_xor"]]
            assert int(np.bitwise_xor.reduce(hs)) == int(c["hash_xor"], 16)
            assert [int(x) for x in kmers[:16].reshape(-1)] == [int(x, 16) for x in c["first16"]]
            assert [int(x) for x in kmers[-16:].reshape(-1)] == [int(x, 16) for x in c["last16"]]
            assert [int(x) for x in hs[:16]] == [int(x, 16) for x in c["first16_hashes"]]
            assert [int(x) for x in hs[-16:]] == [int(x, 16) for x in c["last16_hashes"]]
        else:
            buf = synth(ctx, int(c["seed"], 16), 0, L // 16 + 1, 4, c["ambig_per_65536"])
            seq = cap.Seq(buf.data_ptr(), L, 0, 0, 4, 0)
            for stride, n_key, xor_key in ((1, "n", "kmer_xor"), (3, "lattice3_n", "lattice3_kmer_xor")):
                km_d, st_d = dev_empty(L), dev_empty(L)
                ctx.check(ctx.lib.kmers_unambiguous(ctx.handle, C.byref(seq), K, stride, km_d.data_ptr(), st_d.data_ptr(), L,
                                                    cap.MEM_DEVICE, C.byref(res)), "unambiguous")
                m = int(res.n_out)
                assert m == c[n_key]
                assert int(np.bitwise_xor.reduce(host_u64(km_d)[:m])) == int(c[xor_key], 16)
                if stride == 1:
                    st = st_d.cpu().numpy()[:m]
                    assert int(st.sum()) == c["start_sum"] and [int(x) for x in st[:16]] == c["first16_starts"]
            out = dev_empty(L)
            rc = ctx.lib.kmers_spaced(ctx.handle, C.byref(seq), K, 3, 2, out.data_ptr(), cap.MEM_DEVICE, C.byref(res))
            assert rc == 1 and (res.err_pos, res.err_enc) == (c["strict_spaced_error"]["pos"], c["strict_spaced_error"]["enc"])


def test_batch_of_reads_over_the_c2_pool_full_size(km, ctx):
    """8 M reads x 125 bases cut from the 1 Gbase C2 pool: read i, element j must equal window
    125 i + j of the whole-pool CanonicalDNAMers{31} + fx_hash pass (no oracle at this size)."""
    import ctypes as C
    cap = km._capi
    L, K, bits, R = 1_000_000_000, 31, 4, 125
    n_reads = L // R
    per = R - K + 1
    nw = (L * bits + 63) // 64
    buf = synth(ctx, GOLDEN ^ 2, 0, nw, bits)
    seq = cap.Seq(buf.data_ptr(), L, 0, 0, bits, 0)
    res = cap.Result()
    n = L - K + 1
    whole_k, whole_h = dev_empty(n), dev_empty(n)
    ctx.check(ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, whole_k.data_ptr(), whole_h.data_ptr(), 7, cap.MEM_DEVICE,
                                      C.byref(res)), "canonical")
    spans = torch.stack([torch.arange(n_reads, dtype=torch.int64, device="cuda:0") * R,
                         torch.full((n_reads,), R, dtype=torch.int64, device="cuda:0")], dim=1).contiguous()
    total = n_reads * per
    out_k, out_h = dev_empty(total), dev_empty(total)
    torch.cuda.synchronize()
    rc = ctx.lib.kmers_batch(ctx.handle, C.byref(seq), spans.data_ptr(), n_reads, cap.BATCH_CANONICAL, K, 2, out_k.data_ptr(),
                             out_h.data_ptr(), 7, None, total, cap.MEM_DEVICE | cap.SPANS_DEVICE, C.byref(res))
    assert rc == 0 and res.n_out == total, ctx.last_error()
    step = 1 << 20  # reads per comparison chunk
    for lo in range(0, n_reads, step):
        hi = min(n_reads, lo + step)
        # windows [125 i, 125 i + 95) of the whole-pool arrays, as a strided view
        wk = torch.as_strided(whole_k, (hi - lo, per), (R, 1), lo * R)
        wh = torch.as_strided(whole_h, (hi - lo, per), (R, 1), lo * R)
        assert torch.equal(out_k[lo * per:hi * per].view(hi - lo, per), wk)
        assert torch.equal(out_h[lo * per:hi * per].view(hi - lo, per), wh)
