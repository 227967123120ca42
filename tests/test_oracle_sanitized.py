"""The C oracle under UndefinedBehaviorSanitizer (CPU only): the Julia semantics it restates
(shift counts masked to 6 bits, `1 << 64 == 0` in get_mask, trailing_zeros(0) == 64) are exactly
where a C translation invites UB.  Runs the golden and boundary cases in a
subprocess against a -fsanitize=undefined -fno-sanitize-recover build; any UB aborts it."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_ub_free(tmp_path):
    lib = tmp_path / "libkmers_oracle_san.so"
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fPIC", "-fsanitize=undefined", "-fno-sanitize-recover=all",
                    "-shared", "-o", str(lib), os.path.join(ROOT, "oracle", "kmers_oracle.c")], check=True)
    script = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
        import numpy as np
        import naive
        from oracle import pyoracle
        orc = pyoracle.Oracle({str(lib)!r})
        rng = np.random.default_rng(0)
        for K in (1, 16, 31, 32, 33, 63, 64, 65, 96, 128):
            for src in (2, 4):
                for dst in (2, 4):
                    if orc.nwords(K, dst) > 8:
                        continue
                    text = naive.random_text(rng, K + 70)
                    seq = naive.longseq_words(text, src)
                    orc.fwrv(seq, len(text), src, dst, K)
                    orc.canonical(seq, len(text), src, dst, K, seed=2**64 - 1)
                    orc.spaced(seq, len(text), src, dst, K, 3)
                    orc.spaced(seq, len(text), src, dst, K, K + 5)
                    w = naive.kmer_words(text[:K], dst)
                    orc.reverse_complement(w, K, dst); orc.canonical_kmer(w, K, dst); orc.iscanonical(w, K, dst)
                    orc.longseq_from_kmer(w, K, dst); orc.kmer_from_longseq(naive.longseq_words(text[:K], dst), K, K, dst)
                    if K * dst <= 128:
                        v, _ = orc.as_integer(w, K, dst); orc.from_integer(v, K, dst)
            amb = naive.random_text(rng, K + 200, p_amb=0.2)
            if orc.nwords(K, 2) <= 8:
                orc.unambiguous(naive.longseq_words(amb, 4), len(amb), 4, K)      # trailing_zeros(0) == 64 path
                orc.unambiguous(naive.ascii_words(amb), len(amb), 8, K)
                orc.fw_kmers(naive.longseq_words(amb, 4), len(amb), 4, 2, K)       # EncodeError path
                orc.minimizers(naive.longseq_words(naive.random_text(rng, K + 60), 2), K + 60, 2, 2, K, 9, 4, 0)
        orc.synth_words(2**64 - 1, 2**40, 64, 4, 2621)
        print("UBSAN-CLEAN")
    """)
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "UBSAN-CLEAN" in out.stdout, out.stderr[-2000:]


def test_oracle_under_address_sanitizer(tmp_path):
    """tests/c/oracle_asan_main.c: every iterator and element-wise function of the oracle on exact-size heap
    buffers (AddressSanitizer + UBSan): no access outside ceil(len * bps / 64) source words or n * N output
    words.  (GPU AddressSanitizer is unavailable on this pool; the kernels' own reads are covered by the parity
    tests on exact-size device buffers, tests/test_gpu_parity.py.)"""
    exe = tmp_path / "orc_asan"
    subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe),
                    os.path.join(ROOT, "tests", "c", "oracle_asan_main.c"), os.path.join(ROOT, "oracle", "kmers_oracle.c")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no invalid access" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
