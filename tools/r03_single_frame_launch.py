"""One 4K frame per launch, back to back on one stream over a ring of 48 distinct frames (nothing stays cached): time per
launch of hsvfilter against the grid cap (MI355_FLAG_HSV_BLOCKS_PER_CU), and of the colorlut launch behind it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
W, H, N = 3840, 2160, 48
ctx = mi355fx.Context(0)
one = synth.smooth_frame(W, H)
fb = one.nbytes
d = ctx.alloc(fb * N); dd = ctx.alloc(fb * N)
for i in range(N):
    ctx.h2d(d + i * fb, np.roll(one, 4 * 17 * i, axis=1).reshape(-1))
st = synth.HSV_SETTINGS["hue90"]
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
def run(pair, reps=6):
    best = 1e9
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(N):
            ctx.hsvfilter_frames_device(d + i * fb, 1, fb, W, H, W * 4, "RGBA", st)
            if pair: ctx.colorlut_frames_device(d + i * fb, fb, W * 4, dd + i * fb, fb, W * 4, 1, W, H, "RGBA")
        ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / N)
    return best
mi355fx.warm_clocks(lambda: ctx.hsvfilter_frames_device(d, 8, fb, W, H, W * 4, "RGBA", st), ctx.synchronize)
for _ in range(40): run(True, 1)      # lets the colorlut auto-choice settle on its table
def run_f(F, reps=6):
    best = 1e9
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter()
        for i in range(0, N, F):
            ctx.hsvfilter_frames_device(d + i * fb, F, fb, W, H, W * 4, "RGBA", st)
        ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / (N // F))
    return best
for F in (1, 2, 4, 8):
    row = []
    for bpc in (64, 32, 16, 8):
        ctx.set_flag(mi355fx.FLAG_HSV_BLOCKS_PER_CU, bpc)
        row.append("cap %d: %.2f us" % (bpc, run_f(F) * 1e6))
    print("%d frame(s) per launch: " % F + ", ".join(row), flush=True)
for bpc in (64, 32, 16, 8, 4):
    ctx.set_flag(mi355fx.FLAG_HSV_BLOCKS_PER_CU, bpc)
    a = run(False); b = run(True)
    print("blocks/CU cap %3d: hsvfilter %.2f us per 1-frame launch (%.1f %% of 8 TB/s), pair %.2f us -> %.0f frames/s" % (bpc, a * 1e6, 2 * fb / a / 8e10, b * 1e6, 1 / b), flush=True)
