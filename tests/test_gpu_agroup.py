"""GPU tests for the audio dispatcher (mi355_agroup_*): independent rsaudioecho / ebur128level / audioloudnorm instances that share
launch sets (audio/audiofx/src/audioecho/imp.rs:205-227, ebur128level/imp.rs:682-745, audioloudnorm/imp.rs:1545-1586: one
instance per stream, one buffer per call). The bar: every member's samples / meter readings == a single-instance context fed
the same buffers, bit for bit (and against the C oracle where the single-instance tests compare against it); ragged arrival
orders, threads, ragged buffer sizes (echo), detach, the lock-step timeout, destroy."""
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- rsaudioecho

def test_echo_members_equal_single_instances_with_ragged_everything(mi355lib, oracle):
    """Six members with their own parameters, sample types and BUFFER SIZES per interval, submitted in shuffled orders: every
    member's samples and ring == its own single-instance context (which tests/test_gpu_elements.py pins against the oracle; the
    next test compares with the oracle directly)."""
    import mi355fx
    rate, ch = 48000, 2
    ring = rate * ch
    n_m = 6
    rng = np.random.default_rng(1)
    par = [(24000, 0.6, 0.4), (0, 0.5, 0.0), (4800, 0.3, 0.9), (ring, 1.0, 0.5), (7, 0.25, 0.25), (96, 0.8, 0.0)]
    dt = [np.float32, np.float64, np.float32, np.float64, np.float32, np.float32]
    g = mi355fx.AudioGroup("echo", n_m, ring_len=ring)
    singles = [mi355fx.Context(0) for _ in range(n_m)]
    for c in singles:
        c.echo_setup(ring)
    assert oracle.Echo(10 ** 9, rate, ch).ring_len == ring
    try:
        for it in range(12):
            sizes = [int(rng.integers(1, 3000)) * ch if it % 3 else 960 for _ in range(n_m)]
            bufs = [rng.standard_normal(sizes[m]).astype(dt[m]) for m in range(n_m)]
            exp = [singles[m].echo_process(bufs[m].copy(), par[m][0], par[m][1], par[m][2]) for m in range(n_m)]
            order = rng.permutation(n_m)
            tk = {}
            for m in order:
                tk[m] = g.submit_echo(int(m), bufs[m], *par[m])
            for m in rng.permutation(n_m):
                assert g.wait(tk[m]) == sizes[m]
                assert (bufs[m] == exp[m]).all(), (it, m)
        for m in range(n_m):
            r1, p1 = g.echo_state(m, ring)
            r0, p0 = singles[m].echo_state(ring)
            assert p1 == p0 and (r1 == r0).all()
        bufs_n, sets, largest = g.stats()
        assert bufs_n == 12 * n_m and sets == 12 and largest == n_m     # one launch set per interval for all six
    finally:
        g.close()
        for c in singles:
            c.close()


def test_echo_against_the_oracle_and_device_buffers(mi355lib, oracle):
    import mi355fx
    rate, ch, n = 48000, 2, 960
    g = mi355fx.AudioGroup("echo", 3, ring_len=rate * ch)
    c = mi355fx.Context(0)
    os_ = [oracle.Echo(10 ** 9, rate, ch) for _ in range(3)]
    rng = np.random.default_rng(3)
    try:
        dev = c.alloc(n * 4)
        for it in range(30):
            x = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
            exp = [os_[m].process(x[m].copy(), 250 * 10 ** 6, 0.6, 0.4) for m in range(3)]
            c.h2d(dev, x[2].view(np.uint8))
            c.synchronize()
            t0 = g.submit_echo(0, x[0], 24000, 0.6, 0.4)
            t2 = g.submit_echo(2, dev, 24000, 0.6, 0.4, n=n, is_f64=False)     # a device-resident member
            t1 = g.submit_echo(1, x[1], 24000, 0.6, 0.4)
            for t in (t1, t0, t2):
                g.wait(t)
            back = np.zeros(n, np.float32)
            c.d2h(back.view(np.uint8), dev)
            assert (x[0] == exp[0]).all() and (x[1] == exp[1]).all() and (back == exp[2]).all(), it
        c.free(dev)
    finally:
        g.close()
        c.close()


def test_echo_threads_linger_and_a_missing_member(mi355lib):
    """Eight element threads submit + wait at once: one launch set per interval. A member that stops submitting does not hold the
    others: after the linger the waiter launches whoever is there."""
    import mi355fx
    n_m, rounds, n = 8, 20, 960
    ring = 96000
    g = mi355fx.AudioGroup("echo", n_m, ring_len=ring)
    g.set_linger(2_000_000)
    singles = [mi355fx.Context(0) for _ in range(n_m)]
    for c in singles:
        c.echo_setup(ring)
    rng = np.random.default_rng(4)
    data = [[rng.standard_normal(n).astype(np.float32) for _ in range(rounds)] for _ in range(n_m)]
    exp = [[singles[m].echo_process(data[m][r].copy(), 24000, 0.6, 0.4) for r in range(rounds)] for m in range(n_m)]
    ok = [True] * n_m
    try:
        def element(m):
            for r in range(rounds):
                buf = data[m][r].copy()
                g.wait(g.submit_echo(m, buf, 24000, 0.6, 0.4))
                ok[m] &= bool((buf == exp[m][r]).all())

        ts = [threading.Thread(target=element, args=(m,)) for m in range(n_m)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert all(ok)
        assert g.stats() == (n_m * rounds, rounds, n_m)
        # member 7 is silent from now on (not detached): the others go on after a 10 ms linger
        g.set_linger(10_000)
        buf = [rng.standard_normal(n).astype(np.float32) for _ in range(n_m - 1)]
        e2 = [singles[m].echo_process(buf[m].copy(), 24000, 0.6, 0.4) for m in range(n_m - 1)]
        t0 = time.perf_counter()
        tk = [g.submit_echo(m, buf[m], 24000, 0.6, 0.4) for m in range(n_m - 1)]
        for t in tk:
            g.wait(t)
        assert 0.008 < time.perf_counter() - t0 < 1.0
        assert all((buf[m] == e2[m]).all() for m in range(n_m - 1))
        # detached, the set is complete without it: no linger
        g.detach(7)
        buf = [rng.standard_normal(n).astype(np.float32) for _ in range(n_m - 1)]
        e3 = [singles[m].echo_process(buf[m].copy(), 24000, 0.6, 0.4) for m in range(n_m - 1)]
        g.set_linger(5_000_000)
        t0 = time.perf_counter()
        tk = [g.submit_echo(m, buf[m], 24000, 0.6, 0.4) for m in range(n_m - 1)]
        for t in tk:
            g.wait(t)
        assert time.perf_counter() - t0 < 1.0
        assert all((buf[m] == e3[m]).all() for m in range(n_m - 1))
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_echo(7, buf[0], 24000, 0.6, 0.4)
    finally:
        g.close()
        for c in singles:
            c.close()


def test_echo_errors(mi355lib):
    import mi355fx
    g = mi355fx.AudioGroup("echo", 2, ring_len=100)
    try:
        x = np.zeros(10, np.float32)
        with pytest.raises(mi355fx.Mi355Error) as e:
            g.submit_echo(0, x, 101, 0.5, 0.5)          # RingBufferIter::new: assert!(size >= delay)
        assert e.value.status == mi355fx.ERR_INVALID_ARG
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_echo(2, x, 1, 0.5, 0.5)            # no such member
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_ebur128(0, x)                      # another kind's entry point
        t = g.submit_echo(0, x, 1, 0.5, 0.5)
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_echo(0, x, 1, 0.5, 0.5)            # the previous buffer has not been waited for
        g.wait(t)                                        # (member 1 missing, linger 0: launched at once)
        with pytest.raises(mi355fx.Mi355Error):
            g.wait(0)
    finally:
        g.close()
    with pytest.raises(mi355fx.Mi355Error):
        mi355fx.AudioGroup("echo", 0, ring_len=100)


def test_echo_results_survive_a_growing_slab(mi355lib):
    """A member's result waits in the staging slab until the member collects it. Another member's larger buffer replaces the slab in
    between: what has not been collected moves along (round 6's stress run caught a result read from the slab that had just been
    replaced - one thread is enough to show it)."""
    import mi355fx
    ring = 4096
    g = mi355fx.AudioGroup("echo", 3, ring_len=ring)
    singles = [mi355fx.Context(0) for _ in range(3)]
    for c in singles:
        c.echo_setup(ring)
    rng = np.random.default_rng(3)
    try:
        g.set_linger(0)
        n_big = 600
        for rnd in range(6):
            small = [rng.standard_normal(500).astype(np.float32) for _ in range(2)]
            exp = [singles[m].echo_process(small[m].copy(), 1000, 0.5, 0.3) for m in range(2)]
            t1 = g.submit_echo(1, small[1], 1000, 0.5, 0.3)
            t0 = g.submit_echo(0, small[0], 1000, 0.5, 0.3)
            g.wait(t0)                                   # the launch set of {0, 1} runs; member 1 has not collected its result
            assert (small[0] == exp[0]).all()
            n_big *= 4                                   # ... and member 2's buffer needs larger slots every round
            big = rng.standard_normal(n_big).astype(np.float64)
            exp_big = singles[2].echo_process(big.copy(), 777, 0.25, 0.0)
            t2 = g.submit_echo(2, big, 777, 0.25, 0.0)
            g.wait(t1)
            assert (small[1] == exp[1]).all(), rnd
            g.wait(t2)
            assert (big == exp_big).all(), rnd
    finally:
        g.close()
        for c in singles:
            c.close()


# ---------------------------------------------------------------- ebur128level

@pytest.mark.parametrize("rate,ch,dtype", [(48000, 2, np.float32), (44100, 1, np.int16), (96000, 6, np.float64)])
def test_ebur128_members_equal_single_meters(mi355lib, rate, ch, dtype):
    """Five meters in one group, fed 100 ms / odd-sized buffers from shuffled orders (host and device members): every reading of
    every member == a single-instance meter fed the same buffers (which tests/test_gpu_ebur128.py holds == the oracle)."""
    import mi355fx
    n_m = 5
    rng = np.random.default_rng(rate + ch)
    g = mi355fx.AudioGroup("ebur128", n_m, channels=ch, rate=rate, mode=63)
    singles = [mi355fx.Context(0) for _ in range(n_m)]
    for c in singles:
        c.ebur128_setup(ch, rate, 63)
    dev_ctx = mi355fx.Context(0)
    try:
        t = 0
        for it, frames in enumerate([rate // 10, 1234, rate // 10, 3 * rate // 10, 777, rate, rate // 10, 4001]):
            tt = (t + np.arange(frames)) / rate
            t += frames
            bufs = []
            for m in range(n_m):
                amp = 0.02 * (m + 1) * (1.0 + 0.5 * np.sin(2 * np.pi * 0.7 * tt))
                x = np.stack([amp * np.sin(2 * np.pi * (300.0 + 50 * m + 7 * c) * tt) for c in range(ch)], 1) + 1e-3 * rng.standard_normal((frames, ch))
                if dtype == np.int16:
                    x = np.clip(np.rint(x * 32767 * 4), -32768, 32767).astype(np.int16)
                else:
                    x = x.astype(dtype)
                bufs.append(np.ascontiguousarray(x))
            for m in range(n_m):
                singles[m].ebur128_add_frames(bufs[m].reshape(-1))
            tk = {}
            dptr = None
            for m in rng.permutation(n_m):
                if m == 2 and it % 2:      # member 2 hands over device memory every other interval
                    dptr = dev_ctx.alloc(bufs[m].nbytes)
                    dev_ctx.h2d(dptr, bufs[m].reshape(-1).view(np.uint8))
                    dev_ctx.synchronize()
                    tk[m] = g.submit_ebur128(int(m), dptr, frames, {np.dtype(np.int16): 0, np.dtype(np.float32): 2, np.dtype(np.float64): 3}[np.dtype(dtype)])
                else:
                    tk[m] = g.submit_ebur128(int(m), bufs[m].reshape(-1))
            for m in range(n_m):
                assert g.wait(tk[m]) == frames
            if dptr is not None:
                dev_ctx.free(dptr)
            for m in range(n_m):
                s = singles[m]
                own = [s.ebur128_loudness_momentary(), s.ebur128_loudness_shortterm() if t >= 3 * rate else None, s.ebur128_loudness_global(), s.ebur128_relative_threshold()]
                got = [g.loudness(m, 0), g.loudness(m, 1) if t >= 3 * rate else None, g.loudness(m, 2), g.loudness(m, 3)]
                assert got == own, (it, m, got, own)
                for c in range(ch):
                    assert g.peak(m, c) == s.ebur128_sample_peak(c) and g.peak(m, c, True) == s.ebur128_true_peak(c)
        assert g.stats() == (8 * n_m, 8, n_m)
    finally:
        g.close()
        dev_ctx.close()
        for c in singles:
            c.close()


def test_ebur128_members_are_independent_meters(mi355lib):
    """Every member has its own 100 ms phase: buffer sizes differ between members of one launch set, members sit intervals out
    (the waiter lingers, then whoever is there runs), one member is reset (ebur128level's `reset` action, imp.rs:124-139) while
    the others keep their history. Every reading of every member == a single-instance meter that was fed / reset the same way."""
    import mi355fx
    rate, ch, n_m = 48000, 2, 6
    rng = np.random.default_rng(77)
    g = mi355fx.AudioGroup("ebur128", n_m, channels=ch, rate=rate, mode=63)
    singles = [mi355fx.Context(0) for _ in range(n_m)]
    for c in singles:
        c.ebur128_setup(ch, rate, 63)
    fed = [0] * n_m
    try:
        g.set_linger(0, 0)      # whoever is there runs as soon as somebody waits
        sizes = [4800, 1000, 4801, 9600, 333, 14400, 48000, 2400, 7, 4799]
        for it in range(40):
            who = [m for m in range(n_m) if rng.random() < 0.7] or [int(rng.integers(n_m))]
            tk = {}
            for m in who:
                frames = int(sizes[int(rng.integers(len(sizes)))])
                tt = (fed[m] + np.arange(frames)) / rate
                x = np.stack([0.03 * (m + 1) * np.sin(2 * np.pi * (200.0 + 40 * m + 9 * c) * tt) * (1 + 0.6 * np.sin(2 * np.pi * 0.9 * tt)) for c in range(ch)], 1)
                x = np.ascontiguousarray((x + 1e-3 * rng.standard_normal((frames, ch))).astype(np.float32))
                fed[m] += frames
                singles[m].ebur128_add_frames(x.reshape(-1))
                tk[m] = (g.submit_ebur128(m, x.reshape(-1)), frames)
            for m in who:
                assert g.wait(tk[m][0]) == tk[m][1]
            if it in (11, 23):      # one member starts over; nobody else notices
                victim = it % n_m
                g.ebur128_reset(victim)
                singles[victim].ebur128_reset()
                fed[victim] = 0
            for m in range(n_m):
                s = singles[m]
                own = [s.ebur128_loudness_momentary(), s.ebur128_loudness_shortterm(), s.ebur128_loudness_global(), s.ebur128_relative_threshold(), s.ebur128_loudness_range()]
                got = [g.loudness(m, k) for k in range(5)]
                assert got == own, (it, m, got, own)
                for c in range(ch):
                    assert g.peak(m, c) == s.ebur128_sample_peak(c) and g.peak(m, c, True) == s.ebur128_true_peak(c)
        with pytest.raises(mi355fx.Mi355Error):
            t = g.submit_ebur128(0, np.zeros(4800 * ch, np.float32))
            try:
                g.ebur128_reset(0)          # not with a buffer pending
            finally:
                g.wait(t)
        with pytest.raises(mi355fx.Mi355Error):
            t = g.submit_ebur128(0, np.zeros(4800 * ch, np.float32))
            try:
                g.submit_ebur128(1, np.zeros(4800 * ch, np.int16))     # another sample format in the same launch set
            finally:
                g.wait(t)
    finally:
        g.close()
        for c in singles:
            c.close()


def test_ebur128_linger_collects_the_members_that_come_in_time(mi355lib):
    """Three of four members submit from their own threads within the linger; the fourth is paused: one launch set of three, and the
    paused member's meter has not moved (no silence is fed to it)."""
    import mi355fx
    rate, ch = 48000, 2
    g = mi355fx.AudioGroup("ebur128", 4, channels=ch, rate=rate, mode=63)
    single = mi355fx.Context(0)
    single.ebur128_setup(ch, rate, 63)
    rng = np.random.default_rng(5)
    try:
        g.set_linger(1000000, 0)     # (the fourth member never comes: the first waiter lingers its full second for the other two)
        x = (0.1 * rng.standard_normal((3, 19200, ch))).astype(np.float32)
        done = [None] * 3

        def element(m):
            done[m] = g.wait(g.submit_ebur128(m, x[m].reshape(-1)))

        ts = [threading.Thread(target=element, args=(m,)) for m in range(3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=30)
        assert done == [19200] * 3
        assert g.stats() == (3, 1, 3)
        single.ebur128_add_frames(x[1].reshape(-1))
        assert g.loudness(1, 0) == single.ebur128_loudness_momentary()
        assert g.loudness(3, 0) == -np.inf and g.peak(3, 0) == 0.0      # the paused member: a meter that has heard nothing
    finally:
        g.close()
        single.close()


def test_loudnorm_members_are_independent(mi355lib):
    """audioloudnorm members stand at their own frame type: one starts three seconds into the others' streams (its 3 s first frame
    while they hand over 100 ms frames), one pauses, one ends early; a member that does not come holds nobody (linger). A launch
    set runs one launch sequence per class of members at the same frame type and size. Samples == single-instance contexts."""
    import mi355fx
    ch, n_m = 1, 4
    lengths = [4.3, 3.9, 3.5, 3.2]                     # seconds of 192 kHz audio per member
    xs = [_ln_signal(20 + k, lengths[k], ch) for k in range(n_m)]
    exp = []
    for x in xs:
        c = mi355fx.Context(0)
        c.loudnorm_setup(ch)
        parts = [c.loudnorm_push(x)]
        d = c.loudnorm_drain()
        exp.append(np.concatenate(parts + ([d] if d is not None else [])))
        c.close()
    g = mi355fx.AudioGroup("loudnorm", n_m, channels=ch)
    try:
        g.set_linger(0, 0)
        pos, outs, done = [0] * n_m, [[] for _ in range(n_m)], [False] * n_m

        def step(members):
            """every listed member hands over its next frame (or its final rest) in ONE launch set"""
            tk = {}
            for m in members:
                fs = g.loudnorm_frame_size(m)
                x = xs[m]
                if len(x) - pos[m] >= fs:
                    out = np.zeros((max(fs, 19200), ch))
                    tk[m] = (g.submit_loudnorm(m, x[pos[m]:pos[m] + fs], out), out, fs)
                else:
                    out = np.zeros((31 * 19200, ch))
                    tk[m] = (g.submit_loudnorm(m, x[pos[m]:], out, final_frame=True), out, None)
            for m, (t, out, fs) in tk.items():
                n = g.wait(t)
                outs[m].append(out[:n].reshape(-1).copy())
                if fs is None:
                    done[m] = True
                else:
                    pos[m] += fs

        step([0, 1])                  # members 0 and 1 start together: their 3 s first frames, one class
        for _ in range(3):
            step([0, 1])              # 100 ms frames
        step([0, 1, 2])               # member 2 starts late: its first frame beside the others' inner frames (two classes in one set)
        step([0, 2])                  # member 1 pauses
        step([0, 1, 2, 3])            # member 3 starts, member 1 is back
        while not all(done):
            step([m for m in range(n_m) if not done[m]])      # to the end: the final rests come at different times
        for m in range(n_m):
            got = np.concatenate(outs[m])
            assert got.size == exp[m].size, (m, got.size, exp[m].size)
            assert (got == exp[m]).all(), m
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_loudnorm(0, xs[0][:100], np.zeros((19200, ch)))     # not a whole frame
    finally:
        g.close()


# ---------------------------------------------------------------- audioloudnorm

def _ln_signal(seed, seconds, ch=2):
    t = np.arange(int(seconds * 192000)) / 192000
    rng = np.random.default_rng(seed)
    x = np.stack([0.05 * np.sin(2 * np.pi * (440 + 13 * seed + 3 * c) * t) * (1 + 0.5 * np.sin(2 * np.pi * 0.2 * t)) for c in range(ch)], 1)
    for s in rng.uniform(min(3.2, 0.3 * seconds), seconds - 0.3, 6):
        i = int(s * 192000)
        x[i:i + int(rng.integers(10, 3000))] *= rng.uniform(10, 25)
    return x


@pytest.mark.parametrize("seconds", [4.37, 1.5])
def test_loudnorm_members_equal_single_instances(mi355lib, seconds):
    """Four audioloudnorm elements, each with its own adapter, feeding whole frames from their own threads (the first 3 s frame,
    100 ms frames, the final rest / a stream that ends inside its first 3 s): samples == four single-instance contexts."""
    import mi355fx
    n_m, ch = 4, 2
    xs = [_ln_signal(k, seconds, ch) for k in range(n_m)]
    exp = []
    for x in xs:
        c = mi355fx.Context(0)
        c.loudnorm_setup(ch)
        parts = [c.loudnorm_push(x)]
        d = c.loudnorm_drain()
        if d is not None:
            parts.append(d)
        exp.append(np.concatenate(parts))
        c.close()
    g = mi355fx.AudioGroup("loudnorm", n_m, channels=ch)
    got = [None] * n_m
    err = []

    def element(m):
        try:
            x, outs, pos = xs[m], [], 0
            while True:
                fs = g.loudnorm_frame_size(m)
                if len(x) - pos < fs:
                    break
                out = np.zeros((max(fs, 19200), ch))
                n = g.wait(g.submit_loudnorm(m, x[pos:pos + fs], out))
                outs.append(out[:n].reshape(-1).copy())
                pos += fs
            out = np.zeros((31 * 19200, ch))                    # drain(): the rest (possibly nothing) as the final frame
            n = g.wait(g.submit_loudnorm(m, x[pos:], out, final_frame=True))
            outs.append(out[:n].reshape(-1).copy())
            got[m] = np.concatenate(outs)
        except Exception as e:      # noqa: BLE001
            err.append(e)
            g.detach(m)

    try:
        ts = [threading.Thread(target=element, args=(m,)) for m in range(n_m)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not err, err
        for m in range(n_m):
            assert got[m].size == exp[m].size, (m, got[m].size, exp[m].size)
            assert (got[m] == exp[m]).all(), m
        with pytest.raises(mi355fx.Mi355Error):
            g.submit_loudnorm(0, xs[0][:100], np.zeros((19200, ch)))     # not a whole frame
    finally:
        g.close()


def test_loudnorm_push_and_drain_of_members_are_the_single_instance_calls(mi355lib):
    """The shim's form (gst/gstaudioloudnorm.c with MI355_GROUP_MEMBERS): every element pushes whatever buffers arrive and drains at
    EOS; the adapter lives behind mi355_agroup_loudnorm_push. Three members from three threads, odd buffer sizes, through the
    PROCESS-WIDE group of their configuration: samples == mi355_loudnorm_push / _drain on own contexts."""
    import mi355fx
    n_m, ch, seconds = 3, 2, 3.9
    xs = [_ln_signal(10 + k, seconds, ch) for k in range(n_m)]
    chunks = [123457, 96000, 19200 * 7]
    exp = []
    for k, x in enumerate(xs):
        c = mi355fx.Context(0)
        c.loudnorm_setup(ch, loudness_target=-20.0)
        parts = [c.loudnorm_push(x[i:i + chunks[k]]) for i in range(0, len(x), chunks[k])]
        d = c.loudnorm_drain()
        exp.append(np.concatenate(parts + ([d] if d is not None else [])))
        c.close()
    members = [mi355fx.AudioGroup("loudnorm", n_m, shared=True, channels=ch, loudness_target=-20.0) for _ in range(n_m)]
    assert sorted(g.member for g in members) == [0, 1, 2] and len({g.h for g in members}) == 1     # one group, three memberships
    other = mi355fx.AudioGroup("loudnorm", n_m, shared=True, channels=ch, loudness_target=-21.0)   # another configuration: another group
    assert other.h != members[0].h and other.member == 0
    other.close()
    got = [None] * n_m
    err = []

    def element(k):
        try:
            g, x = members[k], xs[k]
            parts = [g.loudnorm_push(g.member, x[i:i + chunks[k]]) for i in range(0, len(x), chunks[k])]
            d = g.loudnorm_drain(g.member)
            got[k] = np.concatenate(parts + ([d] if d is not None else []))
        except Exception as e:      # noqa: BLE001
            err.append(e)
            members[k].detach(members[k].member)

    ts = [threading.Thread(target=element, args=(k,)) for k in range(n_m)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    try:
        assert not err, err
        for k in range(n_m):
            assert got[k].size == exp[k].size and (got[k] == exp[k]).all(), k
    finally:
        for g in members:
            g.close()          # the last release destroys the group
    again = mi355fx.AudioGroup("loudnorm", n_m, shared=True, channels=ch, loudness_target=-20.0)
    assert again.member == 0   # a fresh group: the old one is gone
    again.close()
