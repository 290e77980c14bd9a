"""GPU tests for the compare queue of the dispatcher (mi355_group_submit_compare / _wait_compare): videocompare's Dssim / Blockhash
pairs of INDEPENDENT element instances in shared launch sequences (video/videofx/src/videocompare/imp.rs:316-350 per element).
The bar: every pair's result == what that element's own entry points give (mi355_dssim_create_image + compare_frames;
hash_frame x 2 + distance) bit for bit, and Blockhash == the C oracle; ragged submission orders, mixed classes in one queue,
shared reference frames, a failing member, a rendezvous with linger, threads, destroy with pairs pending."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame(rng, w, h, ch=4, block=8):
    hb, wb = max(1, h // block), max(1, w // block)
    base = np.kron(rng.integers(0, 256, (hb, wb, ch), dtype=np.uint8), np.ones((block, block, 1), np.uint8))[:h, :w]
    base = np.ascontiguousarray(np.pad(base, ((0, h - base.shape[0]), (0, w - base.shape[1]), (0, 0)), mode="edge")).reshape(h, w * ch)
    if ch == 4:
        base[:, 3::4] = 255
    return base


def _noisy(rng, base, amp, ch=4):
    n = np.clip(base.astype(int) + rng.integers(-amp, amp + 1, base.shape), 0, 255).astype(np.uint8)
    if ch == 4:
        n[:, 3::4] = 255
    return n


def _upload(c, a):
    d = c.alloc(a.nbytes)
    c.h2d(d, a.reshape(-1))
    return d


def _own_dssim(c, d_ref, d_frame, w, h, fmt="RGBA"):
    ch = 4 if fmt == "RGBA" else 3
    x = c.dssim_create_image_device(d_ref, w * ch, w, h, fmt)
    v = c.dssim_compare_frames_device(x, [d_frame], w * ch, w, h, fmt)[0]
    c.dssim_free_image(x)
    return v


@pytest.mark.parametrize("w,h,fmt", [(128, 96, "RGBA"), (322, 246, "RGBA"), (641, 363, "RGB"), (1920, 1080, "RGBA"), (35, 20, "RGB")])
def test_dssim_pairs_through_the_group_are_the_elements_own_results(mi355lib, w, h, fmt):
    import mi355fx
    ch = 4 if fmt == "RGBA" else 3
    rng = np.random.default_rng(w * 7 + h)
    n = 7
    ctxs = [mi355fx.Context(0) for _ in range(n)]
    g = mi355fx.Group(0)
    try:
        pairs = []
        for s, c in enumerate(ctxs):
            a = _frame(rng, w, h, ch)
            b = _noisy(rng, a, 2 + 9 * s, ch) if s != 3 else a.copy()     # stream 3: identical frames -> exactly 0
            pairs.append((_upload(c, a), _upload(c, b)))
        exp = [_own_dssim(c, da, db, w, h, fmt) for c, (da, db) in zip(ctxs, pairs)]
        # ragged submission order; waits in another order; twice (the second round reuses pools, events, pinned blocks)
        for order, worder in (((4, 0, 6, 2, 5, 1, 3), (3, 1, 5, 0, 2, 6, 4)), ((0, 1, 2, 3, 4, 5, 6), (6, 5, 4, 3, 2, 1, 0))):
            tk = {s: g.submit_compare(ctxs[s], pairs[s][0], pairs[s][1], w * ch, w, h, fmt, 5) for s in order}
            got = {s: g.wait_compare(tk[s])[0] for s in worder}
            assert [got[s] for s in range(n)] == exp
        assert exp[3] == 0.0
        st = g.compare_stats()
        assert st[0] == 2 * n and st[1] == 2 and st[2] == n      # 14 pairs in two launch sequences of seven
        for c, (da, db) in zip(ctxs, pairs):
            c.free(da); c.free(db)
    finally:
        g.close()
        for c in ctxs:
            c.close()


def test_dssim_pairs_against_the_restatement_and_shared_reference(mi355lib):
    """One reference frame with three pads (videocompare with four sink pads): the reference is hashed once for the three pairs;
    values against the numpy restatement (1e-9: f64 reduction order), and == the aggregator entry point."""
    import mi355fx
    from oracle import dssim_restate as D
    w, h = 322, 246
    rng = np.random.default_rng(5)
    c = mi355fx.Context(0)
    g = mi355fx.Group(0)
    try:
        a = _frame(rng, w, h)
        mods = [_noisy(rng, a, amp) for amp in (3, 20, 90)]
        da = _upload(c, a)
        dm = [_upload(c, m) for m in mods]
        x = c.dssim_create_image_device(da, w * 4, w, h)
        own = c.dssim_compare_frames_device(x, dm, w * 4, w, h)
        c.dssim_free_image(x)
        tk = [g.submit_compare(c, da, d, w * 4, w, h, "RGBA", 5) for d in dm]
        got = [g.wait_compare(t)[0] for t in tk]
        assert got == list(own)
        oa = D.DssimImage(a, w, h, w * 4, 4)
        for v, m in zip(got, mods):
            assert v == pytest.approx(D.compare(oa, D.DssimImage(m, w, h, w * 4, 4)), rel=1e-9, abs=1e-13)
        c.free(da)
        for d in dm:
            c.free(d)
    finally:
        g.close()
        c.close()


def test_blockhash_pairs_and_mixed_classes_in_one_queue(mi355lib, oracle):
    """Blockhash pairs (RGBA and RGB), Dssim pairs of two sizes, all pending at once: each class gets its own launch sequence."""
    import mi355fx
    rng = np.random.default_rng(11)
    c = mi355fx.Context(0)
    g = mi355fx.Group(0)
    try:
        jobs = []   # (w, h, fmt, algo, a, b, d_a, d_b)
        for w, h, fmt, algo in ((640, 480, "RGBA", 4), (128, 96, "RGBA", 5), (640, 480, "RGBA", 4), (320, 240, "RGB", 4), (322, 246, "RGBA", 5),
                                (128, 96, "RGBA", 5), (640, 480, "RGBA", 4), (320, 240, "RGB", 4)):
            ch = 4 if fmt == "RGBA" else 3
            a = _frame(rng, w, h, ch, block=16)
            b = _noisy(rng, a, 60, ch)
            jobs.append((w, h, fmt, algo, a, b, _upload(c, a), _upload(c, b)))
        tk = [g.submit_compare(c, j[6], j[7], j[0] * (4 if j[2] == "RGBA" else 3), j[0], j[1], j[2], j[3]) for j in jobs]
        for t, (w, h, fmt, algo, a, b, da, db) in reversed(list(zip(tk, jobs))):
            ch = 4 if fmt == "RGBA" else 3
            dist, h0, h1 = g.wait_compare(t)
            if algo == 4:
                e0, e1 = oracle.blockhash(a, w, h, w * ch, ch), oracle.blockhash(b, w, h, w * ch, ch)
                assert (h0, h1) == (e0, e1)
                assert dist == float(bin(e0 ^ e1).count("1"))
                assert h0 == c.videocompare_hash_frame(a, w * ch, w, h, fmt) and h1 == c.videocompare_hash_frame(b, w * ch, w, h, fmt)
            else:
                assert dist == _own_dssim(c, da, db, w, h, fmt)
        assert g.compare_stats()[1] == 4   # blockhash RGBA 640x480 / dssim 128x96 / blockhash RGB 320x240 / dssim 322x246
        for j in jobs:
            c.free(j[6]); c.free(j[7])
    finally:
        g.close()
        c.close()


def test_compare_errors_and_a_result_is_collected_once(mi355lib):
    import mi355fx
    rng = np.random.default_rng(2)
    c = mi355fx.Context(0)
    g = mi355fx.Group(0)
    try:
        w, h = 64, 48
        a = _frame(rng, w, h)
        da = _upload(c, a)
        with pytest.raises(mi355fx.Mi355Error) as e:
            g.submit_compare(c, da, da, w * 4, w, h, "RGBA", 0)            # Mean: not batched
        assert e.value.status == mi355fx.ERR_UNSUPPORTED
        with pytest.raises(mi355fx.Mi355Error) as e:
            g.submit_compare(c, da, da, w * 4 - 1, w, h, "RGBA", 5)        # stride shorter than a row
        assert e.value.status == mi355fx.ERR_INVALID_ARG
        with pytest.raises(mi355fx.Mi355Error) as e:
            g.submit_compare(c, da, da, 60 * 4, 60, 44, "RGBA", 4)         # blockhash on a size that is not 8 x 8 whole blocks
        assert e.value.status == mi355fx.ERR_UNSUPPORTED
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_compare(c, da, da, w * 4, w, h, "BGRA", 5)            # videocompare takes RGB / RGBA
        with pytest.raises(mi355fx.Mi355Error):
            g.wait_compare(12345)
        t = g.submit_compare(c, da, da, w * 4, w, h, "RGBA", 5)
        assert g.wait_compare(t)[0] == 0.0
        with pytest.raises(mi355fx.Mi355Error):
            g.wait_compare(t)                                               # collected
        # the filter tickets and the compare tickets are one sequence: a compare ticket is not a frame's
        g.wait(t)
        c.free(da)
    finally:
        g.close()
        c.close()


def test_rendezvous_threads_fill_one_launch_sequence(mi355lib):
    """Eight 'elements' on eight threads, each submitting its pair and waiting at once (what aggregate() does): with a rendezvous
    of eight the pairs of an interval share ONE launch sequence; a straggler is not waited for longer than the linger."""
    import time
    import mi355fx
    w, h, n, rounds = 322, 246, 8, 5
    rng = np.random.default_rng(8)
    ctxs = [mi355fx.Context(0) for _ in range(n)]
    g = mi355fx.Group(0)
    g.set_rendezvous(n, 2_000_000)
    try:
        pairs, exp = [], []
        for s, c in enumerate(ctxs):
            a = _frame(rng, w, h)
            b = _noisy(rng, a, 5 + 7 * s)
            pairs.append((_upload(c, a), _upload(c, b)))
            exp.append(_own_dssim(c, pairs[-1][0], pairs[-1][1], w, h))
        got = [[None] * rounds for _ in range(n)]
        bar = threading.Barrier(n)

        def element(s):
            for r in range(rounds):
                bar.wait()
                if s == 5:
                    time.sleep(0.01 * r)       # ragged arrival
                t = g.submit_compare(ctxs[s], pairs[s][0], pairs[s][1], w * 4, w, h, "RGBA", 5)
                got[s][r] = g.wait_compare(t)[0]

        ts = [threading.Thread(target=element, args=(s,)) for s in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for s in range(n):
            assert got[s] == [exp[s]] * rounds
        pairs_n, seqs, largest = g.compare_stats()
        assert pairs_n == n * rounds and seqs == rounds and largest == n
        # a straggler that never comes: the waiter launches alone after the linger
        g.set_rendezvous(n, 20_000)
        t0 = time.perf_counter()
        t = g.submit_compare(ctxs[0], pairs[0][0], pairs[0][1], w * 4, w, h, "RGBA", 5)
        assert g.wait_compare(t)[0] == exp[0]
        assert 0.015 < time.perf_counter() - t0 < 1.0
        for c, (da, db) in zip(ctxs, pairs):
            c.free(da); c.free(db)
    finally:
        g.close()
        for c in ctxs:
            c.close()


def test_destroy_with_pairs_pending_and_a_bad_pointer_in_a_batch(mi355lib):
    """Pairs still pending at destroy are launched and waited for (their frames are the callers'); then a group whose batch carries
    a frame the kernels cannot read must report to THAT batch's waiters and stay usable is not testable without faulting the GPU -
    so the failing-member case here is the refused submit (nothing queued, the others unaffected)."""
    import mi355fx
    rng = np.random.default_rng(3)
    c = mi355fx.Context(0)
    g = mi355fx.Group(0)
    w, h = 128, 96
    a = _frame(rng, w, h)
    b = _noisy(rng, a, 30)
    da, db = _upload(c, a), _upload(c, b)
    exp = _own_dssim(c, da, db, w, h)
    t1 = g.submit_compare(c, da, db, w * 4, w, h, "RGBA", 5)
    with pytest.raises(mi355fx.Mi355Error):
        g.submit_compare(c, da, 0, w * 4, w, h, "RGBA", 5)         # the failing member: refused, not queued
    t2 = g.submit_compare(c, da, db, w * 4, w, h, "RGBA", 5)
    assert g.wait_compare(t2)[0] == exp and g.wait_compare(t1)[0] == exp
    g.submit_compare(c, da, db, w * 4, w, h, "RGBA", 5)             # never waited for
    g.submit_compare(c, db, da, w * 4, w, h, "RGBA", 4)
    g.close()                                                        # launches, waits, frees
    assert _own_dssim(c, da, db, w, h) == exp                        # the device is fine, the frames were only read
    c.free(da); c.free(db)
    c.close()
