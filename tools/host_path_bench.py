#!/usr/bin/env python3
"""tools/host_path_bench.py — PCIe-inclusive rate of the host entry points (what an unmodified pipeline with
system-memory GstBuffers pays): hsvfilter_frame_ip + colorlut_frame on one 4K RGBA frame per call."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H = 3840, 2160
ctx = mi355fx.Context(0)
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
frame = synth.smooth_frame(W, H).reshape(-1).copy()
out = np.zeros_like(frame)
st = synth.HSV_SETTINGS["hue90"]
for _ in range(3):
    ctx.hsvfilter_frame_ip(frame, W, W * 4, "RGBA", st)
    ctx.colorlut_frame(frame, W * 4, out, W * 4, W, H, "RGBA")
n = 30
t0 = time.perf_counter()
for _ in range(n):
    ctx.hsvfilter_frame_ip(frame, W, W * 4, "RGBA", st)
t1 = time.perf_counter()
for _ in range(n):
    ctx.colorlut_frame(frame, W * 4, out, W * 4, W, H, "RGBA")
t2 = time.perf_counter()
mb = frame.nbytes / 1e6
print("hsvfilter_frame_ip : %.2f ms/frame (%.1f GB/s over PCIe, H2D+D2H)" % ((t1 - t0) / n * 1e3, 2 * mb / ((t1 - t0) / n * 1e3)))
print("colorlut_frame     : %.2f ms/frame (%.1f GB/s over PCIe, H2D+D2H)" % ((t2 - t1) / n * 1e3, 2 * mb / ((t2 - t1) / n * 1e3)))
print("chain host path    : %.1f frames/s (pageable numpy buffers)" % (n / ((t2 - t0) / 2) / 2 * 1))
print("chain host path    : %.1f frames/s" % (1.0 / ((t1 - t0) / n + (t2 - t1) / n)))
