#!/usr/bin/env python3
"""tools/bench_elements.py — per-element device-resident kernel rates for the non-headline configs of
BASELINE.json and the secondary kernels (wall clock around N async launches + one synchronize)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import FMT, HsvDetectSettings, synth
from mi355fx.cube import parse_cube

ctx = mi355fx.Context(0)


def timeit(fn, n=200, warm=3):
    mi355fx.warm_clocks(fn, ctx.synchronize)
    for _ in range(warm):
        fn()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n


def report(name, secs, nbytes, frames):
    print("%-58s %8.3f ms  %7.0f GB/s (%.1f%% of 8 TB/s)  %9.0f frames/s" % (name, secs * 1e3, nbytes / secs / 1e9, nbytes / secs / 8e12 * 100, frames / secs))


# config 2: hsvfilter hue-shift on 1920x1080 BGRx, batch 32
W, H, B = 1920, 1080, 32
fr = np.stack([synth.smooth_frame(W, H)] * B)
d = ctx.alloc(fr.nbytes); ctx.h2d(d, fr)
st = synth.HSV_SETTINGS["hue90"]
report("hsvfilter 1080p BGRx hue-shift=90 (config 2), batch 32", timeit(lambda: ctx.hsvfilter_frames_device(d, B, W * H * 4, W, H, W * 4, "BGRx", st)), 2 * fr.nbytes, B)
report("hsvfilter 1080p BGRx mixed settings, batch 32", timeit(lambda: ctx.hsvfilter_frames_device(d, B, W * H * 4, W, H, W * 4, "BGRx", synth.HSV_SETTINGS["mixed"])), 2 * fr.nbytes, B)
mi355fx_flag = mi355fx.FLAG_FORCE_GENERIC
ctx.set_flag(mi355fx_flag, 1)
report("hsvfilter 1080p BGRx GENERIC (literal) kernel, batch 32", timeit(lambda: ctx.hsvfilter_frames_device(d, B, W * H * 4, W, H, W * 4, "BGRx", st), n=5), 2 * fr.nbytes, B)
ctx.set_flag(mi355fx_flag, 0)
# wide hue shifts (360 < |shift| <= 2^22: FAST since round 3) and a non-finite shift (memoised table)
report("hsvfilter 1080p BGRx hue-shift=725.5 (wide), batch 32", timeit(lambda: ctx.hsvfilter_frames_device(d, B, W * H * 4, W, H, W * 4, "BGRx", (725.5, 1.0, 0.0, 1.0, 0.0))), 2 * fr.nbytes, B)
ctx.free(d)
# 3-byte formats: natural-like RGB frames (the RGBA frame without its alpha byte), 8 x 4K
W3, H3 = 3840, 2160
one = synth.smooth_frame(W3, H3).reshape(H3, W3, 4)
fr3 = np.ascontiguousarray(np.stack([one[..., :3]] * 8))
d3 = ctx.alloc(fr3.nbytes); ctx.h2d(d3, fr3)
for f3 in ("RGB", "BGR"):
    report("hsvfilter 4K %s (3 B/px, 16-pixel chunks), batch 8" % f3, timeit(lambda: ctx.hsvfilter_frames_device(d3, 8, W3 * H3 * 3, W3, H3, W3 * 3, f3, st), n=50), 2 * fr3.nbytes, 8)
ctx.free(d3)
# padded rows (stride = row + 64 bytes): the strided kernel
pad = W3 * 4 + 64
frp = np.zeros((8, H3, pad), np.uint8); frp[:, :, : W3 * 4] = one.reshape(H3, W3 * 4)
dp = ctx.alloc(frp.nbytes); ctx.h2d(dp, frp)
report("hsvfilter 4K RGBA padded rows (stride + 64 B), batch 8", timeit(lambda: ctx.hsvfilter_frames_device(dp, 8, pad * H3, W3, H3, pad, "RGBA", st), n=50), 2 * 8 * H3 * W3 * 4, 8)
ctx.free(dp)

# hsvdetector 4K RGBx -> RGBA
W, H, B = 3840, 2160, 8
fr = np.stack([synth.smooth_frame(W, H)] * B)
ds, dd = ctx.alloc(fr.nbytes), ctx.alloc(fr.nbytes); ctx.h2d(ds, fr)
s = HsvDetectSettings(120.0, 40.0, 0.8, 0.5, 0.7, 0.6)
L = ctx.L
s240 = HsvDetectSettings(240.0, 40.0, 0.8, 0.5, 0.7, 0.6)
report("hsvdetector 4K RGBx->RGBA hue-ref=240 (negative offset class), batch 8", timeit(lambda: L.mi355_hsvdetect_frames_device(ctx.h, ds, W * H * 4, W * 4, FMT["RGBx"], dd, W * H * 4, W * 4, FMT["RGBA"], B, W, H, C.byref(s240))), 2 * fr.nbytes, B)
report("hsvdetector 4K RGBx->RGBA, batch 8", timeit(lambda: L.mi355_hsvdetect_frames_device(ctx.h, ds, W * H * 4, W * 4, FMT["RGBx"], dd, W * H * 4, W * 4, FMT["RGBA"], B, W, H, C.byref(s))), 2 * fr.nbytes, B)

# colorlut variants on 4K RGBA batch 8
for name, text in (("colorlut 33^3 (config 3), auto kernel choice", synth.cube_text_3d(33)), ("colorlut 17^3, auto kernel choice", synth.cube_text_3d(17)),
                   ("colorlut 65^3, auto kernel choice", synth.cube_text_3d(65)), ("colorlut 1D 1024, auto kernel choice", synth.cube_text_1d(1024))):
    lut = parse_cube(text)
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    report(name + ", 4K RGBA batch 8", timeit(lambda: ctx.colorlut_frames_device(ds, W * H * 4, W * 4, dd, W * H * 4, W * 4, B, W, H, "RGBA"), n=10), 2 * fr.nbytes, B)

# rsaudioecho config 1 (device resident): 10 s stereo 48 kHz f32
x = synth.sine_stereo_f32()
dx = ctx.alloc(x.nbytes); ctx.h2d(dx, x)
ctx.echo_setup(96000)
for fb in (0.4, 0.0):
    secs = timeit(lambda: L.mi355_echo_process_device(ctx.h, dx, x.size, 0, 24000, 0.6, fb), n=20)
    print("%-58s %8.3f ms  -> %.0fx real time (10 s buffer), %.1f Msamples/s" % ("rsaudioecho config 1 feedback=%.1f" % fb, secs * 1e3, 10.0 / secs, x.size / secs / 1e6))
