"""Device against device: mi355_dssim_compare_frames (hash and compare in one pass) must return the f64 bits of
mi355_dssim_create_image + mi355_dssim_compare on random geometries, formats, strides and contents.
Run on the GPU box: python tools/stress_dssim.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = mi355fx.Context(0)
    bad = 0
    for case in range(cases):
        big = rng.integers(0, 8) == 0
        w = int(rng.integers(1, 1400 if big else 300))
        h = int(rng.integers(1, 900 if big else 200))
        fmt = "RGBA" if rng.integers(0, 2) else "RGB"
        ch = 4 if fmt == "RGBA" else 3
        stride = w * ch + int(rng.choice([0, 0, 3, 4, 13, 64]))
        n = int(rng.integers(1, 6))

        def frame():
            f = rng.integers(0, 256, (h, stride), dtype=np.uint8)
            kind = rng.integers(0, 3)
            if kind == 1:
                f[:, : w * ch] = np.clip(np.linspace(0, 255, w * ch)[None, :] + rng.normal(0, 4, (h, w * ch)), 0, 255).astype(np.uint8)
            elif kind == 2:
                f[:, : w * ch] = rng.integers(0, 256)
            if ch == 4 and rng.integers(0, 3):
                f[:, 3: w * 4: 4] = 255
            return f
        ref, others = frame(), [frame() for _ in range(n)]
        if rng.integers(0, 4) == 0:
            others[0] = ref.copy()
        a = ctx.dssim_create_image(ref, stride, w, h, fmt)
        two = []
        for f in others:
            b = ctx.dssim_create_image(f, stride, w, h, fmt)
            two.append(ctx.dssim_compare(a, b))
            ctx.dssim_free_image(b)
        fused = ctx.dssim_compare_frames(a, others, stride, w, h, fmt)
        ctx.dssim_free_image(a)
        if fused != two:
            bad += 1
            print("MISMATCH case", case, w, h, fmt, stride, fused, two, flush=True)
    print("%d cases, %d mismatches" % (cases, bad))


if __name__ == "__main__":
    main()
