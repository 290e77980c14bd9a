"""Differential stress test on the GPU: for random LUTs (3D sizes 2..65, 1D), random hsv settings, random frame geometry
(widths that are / are not multiples of 4, 128, 256; odd heights; several frames per launch) the memoised-table kernels
(colorlut variants 5 and 4, the composed table, the hsvfilter table) must produce exactly what the kernels that build them
produce (variant 6 / arithmetic hsvfilter). Device vs device, so thousands of cases run in minutes.
Run on the GPU box: python tools/stress_tables.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx


def rand_settings(rng):
    k = rng.integers(0, 6)
    hue = [0.0, float(rng.uniform(-360, 360)), float(rng.choice([360.0, -360.0, 1e-30, 180.0])), float(rng.choice([725.5, -1e6, np.inf, np.nan])),
           float(np.float32(rng.normal(0, 120))), 90.0][k]
    def sv():
        j = rng.integers(0, 4)
        return [(1.0, 0.0), (float(rng.uniform(0, 3)), float(rng.uniform(-1, 1))), (float(rng.choice([0.0, -1.0, np.inf, np.nan])), 0.25),
                (float(np.float32(rng.normal(1, 0.5))), float(np.float32(rng.normal(0, 0.2))))][j]
    sm, so = sv(); vm, vo = sv()
    return (hue, sm, so, vm, vo)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = mi355fx.Context(0)
    bad = 0
    for it in range(cases):
        # LUT
        if rng.integers(0, 5) == 0:
            size = int(rng.choice([2, 64, 256, 1024, 4096]))
            table = rng.uniform(-0.1, 1.1, (3, size)).astype(np.float32)
            is3d = False
        else:
            size = int(rng.choice([2, 3, 5, 9, 16, 17, 18, 25, 32, 33, 34, 40, 65]))
            table = rng.uniform(-0.1, 1.1, (size ** 3, 4)).astype(np.float32); table[:, 3] = 1.0
            is3d = True
        lo = rng.uniform(-0.2, 0.1, 3).astype(np.float32); hi = (lo + rng.uniform(0.6, 1.4, 3)).astype(np.float32)
        scale = (np.float32(1.0) / (hi - lo)).astype(np.float32); offset = (-lo * scale).astype(np.float32)
        ctx.colorlut_load(is3d, size, table, scale, offset)
        # geometry
        w = int(rng.choice([4, 60, 124, 128, 132, 256, 260, 500, 512, 640, 1000, 1920, 37, 333]))
        h = int(rng.integers(1, 40))
        n = int(rng.integers(1, 4))
        amp = int(rng.choice([0, 3, 255]))
        base = rng.integers(0, 256, (1, 1, 4), dtype=np.int16) if amp < 255 else np.zeros((1, 1, 4), np.int16)
        frames = np.clip(base + rng.integers(-amp, amp + 1, (n * h, w, 4), dtype=np.int16) + (rng.integers(0, 256, (n * h, w, 4), dtype=np.int16) if amp == 255 else 0), 0, 255).astype(np.uint8).reshape(-1)
        nb = frames.nbytes
        d_s, d_o = ctx.alloc(nb), ctx.alloc(nb)
        ctx.h2d(d_s, frames)
        st = rand_settings(rng)
        outs = {}
        for v in (6, 5, 4):
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
            ctx.colorlut_frames_device(d_s, w * h * 4, w * 4, d_o, w * h * 4, w * 4, n, w, h, "RGBA")
            ctx.synchronize()
            o = np.empty_like(frames); ctx.d2h(o, d_o); outs["lut%d" % v] = o
            ctx.hsv_colorlut_frames_device(d_s, w * h * 4, w * 4, d_o, w * h * 4, w * 4, n, w, h, st)
            ctx.synchronize()
            o = np.empty_like(frames); ctx.d2h(o, d_o); outs["fused%d" % v] = o
        fmt = ["RGBA", "BGRx"][it % 2]
        for mode in (3, 2):
            ctx.set_flag(mi355fx.FLAG_HSV_TABLE, mode)
            ctx.h2d(d_o, frames)
            ctx.hsvfilter_frames_device(d_o, n, w * h * 4, w, h, w * 4, fmt, st)
            ctx.synchronize()
            o = np.empty_like(frames); ctx.d2h(o, d_o); outs["hsv%d" % mode] = o
        ctx.free(d_s); ctx.free(d_o)
        ok = (outs["lut5"] == outs["lut6"]).all() and (outs["lut4"] == outs["lut6"]).all() and (outs["fused5"] == outs["fused6"]).all() and \
             (outs["fused4"] == outs["fused6"]).all() and (outs["hsv2"] == outs["hsv3"]).all()
        if not ok:
            bad += 1
            print("MISMATCH case", it, dict(is3d=is3d, size=size, w=w, h=h, n=n, amp=amp, st=st, fmt=fmt), flush=True)
    print("%d cases, %d mismatches" % (cases, bad))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
