"""Threaded stress of the batching dispatcher (mi355_group_*): T host threads, one stream (context, LUT, buffers) each, submit
frames one frame deep - submit n, wait n-1 - with random pauses, random hsv settings out of a small pool (so that batches form
and split), a few streams of another size or with padded rows (not batchable: their context's own path, in order), random
order_after calls and flushes from a disturber thread. Every frame's output must equal what the stream's own two element calls
(mi355_hsvfilter_frames_device + mi355_colorlut_frames_device on a second context with the same LUT) give for that input - device
against device, byte for byte - and every source buffer must end up filtered in place.
Run on the GPU box: python tools/stress_group.py [threads] [frames per thread] [seed]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = int(sys.argv[2]) if len(sys.argv) > 2 else 40
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 1
SETTINGS = [synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["defaults"], (725.5, 1.2, 0.0, 0.9, 0.05), (-40.0, 1.0, 0.1, 1.0, 0.0)]
LUTS = [parse_cube(synth.cube_text_3d(33)), parse_cube(synth.cube_text_3d(17, amp=0.08))]


def main():
    rng = np.random.default_rng(SEED)
    g = mi355fx.Group(0)
    errors, lock = [], threading.Lock()
    stop = threading.Event()

    def stream(t):
        r = np.random.default_rng(SEED * 1000 + t)
        kind = t % 6   # 0-3: 1080p RGBA (batchable), 4: 720p RGBA (batchable with its like), 5: BGRx / padded rows (own path)
        w, h, fmt, pad = (1920, 1080, "RGBA", 0) if kind < 4 else ((1280, 720, "RGBA", 0) if kind == 4 else (1000, 600, "RGBA", 48 if t % 12 == 5 else 16))
        stride = w * 4 + pad
        lut = LUTS[t % 2]
        c, ref = mi355fx.Context(0), mi355fx.Context(0)
        for x in (c, ref):
            x.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        nb = stride * h
        d_src = [c.alloc(nb) for _ in range(2)]; d_dst = [c.alloc(nb) for _ in range(2)]
        r_src, r_dst = ref.alloc(nb), ref.alloc(nb)
        base = synth.smooth_frame(w, h, seed=100 + t).reshape(h, w * 4)
        tickets = [None, None]
        meta = [None, None]

        def check(slot):
            frame, st = meta[slot]
            got, got_src, exp, exp_src = (np.empty(nb, np.uint8) for _ in range(4))
            c.d2h(got, d_dst[slot]); c.d2h(got_src, d_src[slot])
            ref.h2d(r_src, frame)
            ref.hsvfilter_frames_device(r_src, 1, nb, w, h, stride, fmt, st)
            ref.colorlut_frames_device(r_src, nb, stride, r_dst, nb, stride, 1, w, h, fmt)
            ref.synchronize()
            ref.d2h(exp, r_dst); ref.d2h(exp_src, r_src)
            rows = lambda a: a.reshape(h, stride)[:, : w * 4]
            if not (rows(got) == rows(exp)).all() or not (rows(got_src) == rows(exp_src)).all():
                with lock:
                    errors.append("stream %d (%dx%d %s pad %d): frame differs from the two element calls" % (t, w, h, fmt, pad))

        try:
            for n in range(F):
                slot = n % 2
                if tickets[slot] is not None:          # the frame submitted two iterations ago used this slot: wait for it, check it
                    g.wait(tickets[slot]); check(slot)
                frame = np.zeros((h, stride), np.uint8)
                frame[:, : w * 4] = np.roll(base, 4 * int(r.integers(0, w)), axis=1)
                frame = frame.reshape(-1)
                st = SETTINGS[int(r.integers(0, len(SETTINGS)))] if r.random() < 0.3 else SETTINGS[0]
                c.h2d(d_src[slot], frame)              # (synchronous: the upload is done before the submit)
                meta[slot] = (frame, st)
                tickets[slot] = g.submit_chain(c, d_src[slot], d_dst[slot], w, h, stride, fmt, st)
                if r.random() < 0.2:
                    g.order_after(c, tickets[slot])
                if r.random() < 0.3:
                    time.sleep(float(r.random()) * 0.002)
                other = 1 - slot
                if tickets[other] is not None and r.random() < 0.5:   # one frame deep: wait for the previous frame now
                    g.wait(tickets[other]); check(other); tickets[other] = None
            for slot in range(2):
                if tickets[slot] is not None:
                    g.wait(tickets[slot]); check(slot)
        except Exception as e:  # noqa: BLE001
            with lock:
                errors.append("stream %d: %r" % (t, e))
        finally:
            try:
                g.wait_all()   # (nothing of this stream may still be queued when its buffers go)
            except Exception:  # noqa: BLE001
                pass
            for p in d_src + d_dst:
                c.free(p)
            ref.free(r_src); ref.free(r_dst)
            c.close(); ref.close()

    def disturber():
        r = np.random.default_rng(SEED + 77)
        while not stop.is_set():
            time.sleep(float(r.random()) * 0.003)
            try:
                g.flush() if r.random() < 0.7 else g.stats()
            except Exception as e:  # noqa: BLE001
                with lock:
                    errors.append("disturber: %r" % (e,))

    th = [threading.Thread(target=stream, args=(t,)) for t in range(T)]
    d = threading.Thread(target=disturber)
    t0 = time.perf_counter()
    d.start()
    for x in th:
        x.start()
    for x in th:
        x.join()
    stop.set(); d.join()
    frames, batched, single = g.stats()
    g.close()
    print("%d threads x %d frames in %.1f s: %d frames through the group, %d batched launch pairs, %d through their own path; %d errors" %
          (T, F, time.perf_counter() - t0, frames, batched, single, len(errors)))
    for e in errors[:10]:
        print("  ", e)
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
