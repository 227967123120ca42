"""The reference's iterators are immutable values: any number of host threads may iterate at once.  The C ABI
keeps that with one context (one HIP stream, its own staging buffers) per host thread and no global state:
four threads, each with its own context, run every kind of call concurrently (ctypes releases the GIL during a
call) and must all get the results a single thread gets.  Run with -m gpu."""
import ctypes as C
import threading

import numpy as np
import pytest

import naive

pytestmark = pytest.mark.gpu


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


def test_one_context_per_thread(orc):
    import kmers_jl_amd as km
    cap = km._capi
    rng = np.random.default_rng(77)
    L, K = 300_000, 31
    text = naive.random_text(rng, L)
    words = naive.longseq_words(text, 4)
    ek, eh, _ = orc.canonical(words, L, 4, 2, K)
    esk = np.unique(eh)[:500]
    reads = [naive.random_text(rng, int(l)) for l in rng.integers(20, 400, 3000)]
    spans, pieces, pos = [], [], 0
    for t in reads:
        spans.append((pos, len(t)))
        pieces.append(t)
        pos += len(t)
    pool = naive.ascii_words("".join(pieces))
    span_arr = (cap.Span * len(spans))(*[cap.Span(a, b) for a, b in spans])
    eb = np.concatenate([orc.canonical(naive.ascii_words(t), len(t), 8, 2, K)[1] for t in reads if len(t) >= K])
    eoff = np.concatenate([[0], np.cumsum([max(0, len(t) - K + 1) for t in reads])]).astype(np.uint64)
    esks = [np.unique(eb[int(eoff[i]):int(eoff[i + 1])])[:64] for i in range(len(reads))]
    errors = []

    def worker(tid):
        try:
            ctx = km.Context(0)
            res = cap.Result()
            seq = cap.Seq(words.ctypes.data, L, 0, 0, 4, 0)
            pseq = cap.Seq(pool.ctypes.data, pos, 0, 0, 8, 0)
            for it in range(12):
                out_k, out_h = np.zeros(len(ek), np.uint64), np.zeros(len(ek), np.uint64)
                rc = ctx.lib.kmers_canonical(ctx.handle, C.byref(seq), K, 2, vp(out_k), vp(out_h), 0, cap.MEM_HOST, C.byref(res))
                assert rc == 0 and np.array_equal(out_k, ek[:, 0]) and np.array_equal(out_h, eh), (tid, it, "canonical")
                sk = np.zeros(500, np.uint64)
                rc = ctx.lib.kmers_minhash(ctx.handle, C.byref(seq), K, 2, 0, 500, vp(sk), cap.MEM_HOST, C.byref(res))
                assert rc == 0 and np.array_equal(sk[:res.n_out], esk), (tid, it, "minhash")
                a, b = np.zeros(len(eb), np.uint64), np.zeros(len(eb), np.uint64)
                offs = np.zeros(len(reads) + 1, np.uint64)
                rc = ctx.lib.kmers_batch(ctx.handle, C.byref(pseq), span_arr, len(reads), cap.BATCH_CANONICAL, K, 2, vp(a), vp(b), 0, vp(offs),
                                         len(eb), cap.MEM_HOST, C.byref(res))
                assert rc == 0 and np.array_equal(b, eb) and np.array_equal(offs, eoff), (tid, it, "batch")
                sks, cnt = np.zeros((len(reads), 64), np.uint64), np.zeros(len(reads), np.uint64)
                rc = ctx.lib.kmers_minhash_batch(ctx.handle, C.byref(pseq), span_arr, len(reads), K, 2, 0, 64, vp(sks), vp(cnt), cap.MEM_HOST,
                                                 C.byref(res))
                assert rc == 0, (tid, it, "minhash_batch")
                for i in (0, 1, 17, len(reads) - 1, (tid * 131 + it * 17) % len(reads)):
                    assert cnt[i] == len(esks[i]) and np.array_equal(sks[i, :len(esks[i])], esks[i]), (tid, it, i)
            ctx.close()
        except BaseException as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    assert not any(t.is_alive() for t in threads)
