"""Stress of the round-6 dispatchers, device against device (run on the GPU box: python tools/stress_dispatch.py [seconds]):
 (a) the compare queue (mi355_group_submit_compare): T host threads, each an element with its own context, random classes (three
     sizes x RGB / RGBA x Dssim / Blockhash), random pauses, a random rendezvous setting changed while they run, a thread that
     flushes at random - every result == the element's own entry points on a second context;
 (b) the echo group (mi355_agroup_*): T members with random buffer sizes / sample types / parameters per interval, random pauses
     longer than the linger (partial launch sets), host and device buffers - every buffer == the member's own single-instance context;
 (c) an ebur128level group: T meters fed ragged buffer sizes from their own threads with pauses and resets of single members - every
     reading == the member's own single meter;
 (d) audioloudnorm groups: rounds of 6 streams of random length that start at random times and pause (members at different frame types
     in one launch set), through push / drain - every sample == the stream's own single-instance context.
Prints one line per part: cases, mismatches."""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx

SECONDS = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
T = 12


def part_compare():
    sizes = [(640, 360), (322, 246), (1280, 720)]
    g = mi355fx.Group(0)
    stop = time.time() + SECONDS
    bad, done, lock = [0], [0], threading.Lock()

    def element(k):
        rng = np.random.default_rng(100 + k)
        own, ref = mi355fx.Context(0), mi355fx.Context(0)
        while time.time() < stop:
            w, h = sizes[int(rng.integers(0, 3))]
            fmt = "RGBA" if rng.integers(0, 2) else "RGB"
            ch = 4 if fmt == "RGBA" else 3
            algo = 5 if rng.integers(0, 3) else 4
            if algo == 4 and (w % 8 or h % 8):
                algo = 5
            a = rng.integers(0, 256, size=(h, w * ch), dtype=np.uint8)
            a = np.repeat(np.repeat(a[::8, ::8 * ch], 8, 0), 8 * ch, 1)[:h, :w * ch].copy() if rng.integers(0, 2) else a
            b = np.clip(a.astype(int) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
            if ch == 4:
                a[:, 3::4] = 255; b[:, 3::4] = 255
            da, db = own.alloc(a.nbytes), own.alloc(b.nbytes)
            own.h2d(da, a.reshape(-1)); own.h2d(db, b.reshape(-1))
            n_pairs = int(rng.integers(1, 4))
            tk = [g.submit_compare(own, da, db if j == 0 else da, w * ch, w, h, fmt, algo) for j in range(n_pairs)]
            if rng.integers(0, 4) == 0:
                time.sleep(float(rng.uniform(0, 0.003)))
            got = [g.wait_compare(t) for t in tk]
            ra, rb = ref.alloc(a.nbytes), ref.alloc(b.nbytes)
            ref.h2d(ra, a.reshape(-1)); ref.h2d(rb, b.reshape(-1))
            if algo == 5:
                x = ref.dssim_create_image_device(ra, w * ch, w, h, fmt)
                e = [ref.dssim_compare_frames_device(x, [rb if j == 0 else ra], w * ch, w, h, fmt)[0] for j in range(n_pairs)]
                ref.dssim_free_image(x)
                ok = all(got[j][0] == e[j] for j in range(n_pairs))
            else:
                h0 = ref.videocompare_hash_frames_device(ra, a.nbytes, w * ch, 1, w, h, fmt)[0]
                h1 = ref.videocompare_hash_frames_device(rb, b.nbytes, w * ch, 1, w, h, fmt)[0]
                ok = got[0][1:] == (h0, h1) and got[0][0] == bin(h0 ^ h1).count("1") and all(got[j][1:] == (h0, h0) and got[j][0] == 0 for j in range(1, n_pairs))
            for c, p in ((own, da), (own, db), (ref, ra), (ref, rb)):
                c.free(p)
            with lock:
                done[0] += n_pairs
                bad[0] += 0 if ok else 1
        own.close(); ref.close()

    def meddler():
        rng = np.random.default_rng(7)
        while time.time() < stop:
            g.set_rendezvous(int(rng.integers(0, T + 1)), int(rng.integers(0, 3000)))
            if rng.integers(0, 3) == 0:
                g.flush()
            time.sleep(float(rng.uniform(0, 0.01)))

    ts = [threading.Thread(target=element, args=(k,)) for k in range(T)] + [threading.Thread(target=meddler)]
    for t in ts: t.start()
    for t in ts: t.join()
    st = g.compare_stats()
    g.close()
    print("compare queue: %d pairs through %d launch sequences (largest %d), %d elements with a mismatch" % (done[0], st[1], st[2], bad[0]), flush=True)
    return bad[0]


def part_echo():
    ring = 48000
    g = mi355fx.AudioGroup("echo", T, ring_len=ring)
    g.set_linger(300)
    stop = time.time() + SECONDS
    bad, done, lock = [0], [0], threading.Lock()

    def member(m):
        rng = np.random.default_rng(500 + m)
        single, dev = mi355fx.Context(0), mi355fx.Context(0)
        single.echo_setup(ring)
        while time.time() < stop:
            n = int(rng.integers(1, 4000))
            dt = np.float64 if rng.integers(0, 3) == 0 else np.float32
            par = (int(rng.integers(0, ring + 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 0.95)) if rng.integers(0, 3) else 0.0)
            x = rng.standard_normal(n).astype(dt)
            exp = single.echo_process(x.copy(), *par)
            if rng.integers(0, 3) == 0:     # a device-resident buffer
                d = dev.alloc(x.nbytes); dev.h2d(d, x.view(np.uint8)); dev.synchronize()
                g.wait(g.submit_echo(m, d, *par, n=n, is_f64=(dt == np.float64)))
                got = np.zeros(n, dt); dev.d2h(got.view(np.uint8), d); dev.free(d)
            else:
                got = x.copy()
                g.wait(g.submit_echo(m, got, *par))
            if rng.integers(0, 6) == 0:
                time.sleep(float(rng.uniform(0, 0.002)))     # longer than the linger: the others launch without this member
            with lock:
                done[0] += 1
                bad[0] += 0 if (got == exp).all() else 1
        r1, p1 = g.echo_state(m, ring)
        r0, p0 = single.echo_state(ring)
        with lock:
            bad[0] += 0 if (p1 == p0 and (r1 == r0).all()) else 1
        g.detach(m)
        single.close(); dev.close()

    ts = [threading.Thread(target=member, args=(m,)) for m in range(T)]
    for t in ts: t.start()
    for t in ts: t.join()
    st = g.stats()
    g.close()
    print("echo group: %d buffers through %d launch sets (largest %d), %d mismatches (buffers + final rings)" % (done[0], st[1], st[2], bad[0]), flush=True)
    return bad[0]


def part_ebur128():
    rate, ch = 48000, 2
    g = mi355fx.AudioGroup("ebur128", T, channels=ch, rate=rate, mode=63)
    g.set_linger(300)
    stop = time.time() + SECONDS / 2
    bad, done, lock = [0], [0], threading.Lock()

    def member(m):
        rng = np.random.default_rng(900 + m)
        single = mi355fx.Context(0)
        single.ebur128_setup(ch, rate, 63)
        t = 0
        while time.time() < stop:
            n = int(rng.integers(1, 20000))
            tt = (t + np.arange(n)) / rate
            t += n
            x = np.stack([0.05 * (m + 1) * np.sin(2 * np.pi * (250.0 + 31 * m + 5 * c) * tt) for c in range(ch)], 1) + 2e-3 * rng.standard_normal((n, ch))
            x = np.ascontiguousarray(x.astype(np.float32))
            single.ebur128_add_frames(x.reshape(-1))
            assert g.wait(g.submit_ebur128(m, x.reshape(-1))) == n
            got = [g.loudness(m, k) for k in range(5)] + [g.peak(m, c) for c in range(ch)] + [g.peak(m, c, True) for c in range(ch)]
            own = [single.ebur128_loudness_momentary(), single.ebur128_loudness_shortterm(), single.ebur128_loudness_global(), single.ebur128_relative_threshold(),
                   single.ebur128_loudness_range()] + [single.ebur128_sample_peak(c) for c in range(ch)] + [single.ebur128_true_peak(c) for c in range(ch)]
            if rng.integers(0, 40) == 0:
                g.ebur128_reset(m); single.ebur128_reset(); t = 0
            if rng.integers(0, 6) == 0:
                time.sleep(float(rng.uniform(0, 0.002)))
            with lock:
                done[0] += 1
                bad[0] += 0 if got == own else 1
        g.detach(m)
        single.close()

    ts = [threading.Thread(target=member, args=(m,)) for m in range(T)]
    for t in ts: t.start()
    for t in ts: t.join()
    st = g.stats()
    g.close()
    print("ebur128level group: %d buffers through %d launch sets (largest %d), %d buffers with a reading that differs" % (done[0], st[1], st[2], bad[0]), flush=True)
    return bad[0]


def part_loudnorm():
    n_m, ch = 6, 1
    stop = time.time() + SECONDS / 2
    bad, done, sets = 0, 0, 0
    rnd = 0
    while time.time() < stop:
        rnd += 1
        rng = np.random.default_rng(7000 + rnd)
        xs = []
        for k in range(n_m):
            n = int(rng.uniform(0.4, 3.9) * 192000)
            tt = np.arange(n) / 192000
            x = 0.05 * np.sin(2 * np.pi * (300 + 17 * k) * tt) * (1 + 0.5 * np.sin(2 * np.pi * 0.3 * tt))
            for s0 in rng.uniform(0.1 * n, 0.9 * n, 4):
                x[int(s0):int(s0) + int(rng.integers(10, 2000))] *= rng.uniform(8, 20)
            xs.append(x.reshape(-1, ch))
        g = mi355fx.AudioGroup("loudnorm", n_m, channels=ch)
        g.set_linger(500)
        got = [None] * n_m

        def element(k):
            r2 = np.random.default_rng(rnd * 100 + k)
            time.sleep(float(r2.uniform(0, 0.03)))          # streams start at different times
            x, parts, i = xs[k], [], 0
            while i < len(x):
                c = int(r2.integers(1000, 200000))
                parts.append(g.loudnorm_push(k, x[i:i + c]))
                i += c
                if r2.integers(0, 5) == 0:
                    time.sleep(float(r2.uniform(0, 0.003)))
            d = g.loudnorm_drain(k)
            got[k] = np.concatenate(parts + ([d] if d is not None else []))
            g.detach(k)

        ts = [threading.Thread(target=element, args=(k,)) for k in range(n_m)]
        for t in ts: t.start()
        for t in ts: t.join()
        sets += g.stats()[1]
        g.close()
        for k in range(n_m):
            c = mi355fx.Context(0)
            c.loudnorm_setup(ch)
            p = [c.loudnorm_push(xs[k])]
            d = c.loudnorm_drain()
            exp = np.concatenate(p + ([d] if d is not None else []))
            c.close()
            done += 1
            bad += 0 if (got[k] is not None and got[k].size == exp.size and (got[k] == exp).all()) else 1
    print("audioloudnorm groups: %d streams in %d rounds through %d launch sets, %d streams with a sample that differs" % (done, rnd, sets, bad), flush=True)
    return bad


if __name__ == "__main__":
    sys.exit(1 if (part_compare() + part_echo() + part_ebur128() + part_loudnorm()) else 0)
