"""Stress of the round-6 dispatchers, device against device (run on the GPU box: python tools/stress_dispatch.py [seconds]):
 (a) the compare queue (mi355_group_submit_compare): T host threads, each an element with its own context, random classes (three
     sizes x RGB / RGBA x Dssim / Blockhash), random pauses, a random rendezvous setting changed while they run, a thread that
     flushes at random - every result == the element's own entry points on a second context;
 (b) the echo group (mi355_agroup_*): T members with random buffer sizes / sample types / parameters per interval, random pauses
     longer than the linger (partial launch sets), host and device buffers - every buffer == the member's own single-instance context.
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


if __name__ == "__main__":
    sys.exit(1 if (part_compare() + part_echo()) else 0)
