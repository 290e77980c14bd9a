"""State-machine stress of the auto kernel choice: one context in auto mode and one with the interpolating / arithmetic
kernels pinned receive the same random sequence of calls (LUT reloads, settings changes, launch sizes jumping by more than
2x, smooth / noisy content, colorlut / fused / hsvfilter entry points, bursts without synchronisation); every output must
be identical. Run on the GPU box: python tools/stress_auto.py [steps] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    a, b = mi355fx.Context(0), mi355fx.Context(0)
    b.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
    b.set_flag(mi355fx.FLAG_HSV_TABLE, 3)
    a.set_flag(mi355fx.FLAG_HSV_TABLE, 1)
    W, H, NMAX = 1920, 1080, 4
    smooth = np.stack([synth.smooth_frame(W, H, seed=3 + i) for i in range(NMAX)]).reshape(-1)
    noise = np.stack([synth.noise_frame(W, H, seed=9 + i) for i in range(NMAX)]).reshape(-1)
    nb = smooth.nbytes
    bufs = {}
    for c in (a, b):
        bufs[c] = dict(smooth=c.alloc(nb), noise=c.alloc(nb), out=c.alloc(nb), tmp=c.alloc(nb))
        c.h2d(bufs[c]["smooth"], smooth); c.h2d(bufs[c]["noise"], noise)
    settings = [synth.HSV_SETTINGS["hue90"], synth.HSV_SETTINGS["mixed"], (725.5, 1.0, 0.0, 1.0, 0.0), (float("nan"), 1.2, 0.0, 0.9, 0.1)]
    st = settings[0]
    loaded = False
    bad = 0
    pending = 0
    seen_table = [0, 0, 0]
    for it in range(steps):
        op = rng.integers(0, 100)
        if not loaded or it % 700 == 699:
            size = int(rng.choice([5, 17, 33, 40]))
            table = rng.uniform(0, 1, (size ** 3, 4)).astype(np.float32); table[:, 3] = 1.0
            for c in (a, b):
                c.colorlut_load(True, size, table)
            loaded = True
            continue
        if op < 2:
            st = settings[rng.integers(0, len(settings))]
        content = "noise" if rng.integers(0, 4) == 0 else "smooth"
        # mostly the same launch size, sometimes a jump
        n = NMAX if rng.integers(0, 40) else int(rng.choice([1, 2]))
        h = H if rng.integers(0, 40) else int(rng.choice([8, 64, 540]))
        kind = rng.integers(0, 3)
        for c in (a, b):
            src, out, tmp = bufs[c][content], bufs[c]["out"], bufs[c]["tmp"]
            if kind == 0:
                c.colorlut_frames_device(src, W * h * 4, W * 4, out, W * h * 4, W * 4, n, W, h, "RGBA")
            elif kind == 1:
                c.hsv_colorlut_frames_device(src, W * h * 4, W * 4, out, W * h * 4, W * 4, n, W, h, st)
            else:
                c.d2d(tmp, src, n * W * h * 4) if hasattr(c, "d2d") else None
                c.hsvfilter_frames_device(src if not hasattr(c, "d2d") else tmp, n, W * h * 4, W, h, W * 4, "RGBA", st)
        pending += 1
        if kind == 2 and not hasattr(a, "d2d"):
            # in place on the source: both contexts did the same, keep the sources in step by comparing them
            which = "src"
        else:
            which = "out"
        if rng.integers(0, 3) == 0 or kind == 2:   # otherwise: burst, no synchronisation
            outs = []
            for c in (a, b):
                c.synchronize()
                o = np.empty(n * W * h * 4, np.uint8)
                c.d2h(o, bufs[c][content] if which == "src" else bufs[c]["out"])
                outs.append(o)
            if not (outs[0] == outs[1]).all():
                bad += 1
                print("MISMATCH step", it, dict(kind=int(kind), n=n, h=h, content=content, st=st), int((outs[0] != outs[1]).sum()), flush=True)
            pending = 0
            for k in range(3):
                seen_table[k] += int(a.colorlut_kernel_choice(fused=k)[0])
    print("checks with the table kernel in use (colorlut, fused, hsvfilter):", seen_table)
    print("%d steps, %d mismatches; choice colorlut %s fused %s hsv %s" % (steps, bad, a.colorlut_kernel_choice(), a.colorlut_kernel_choice(fused=1), a.colorlut_kernel_choice(fused=2)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
