"""What a barrier-and-synchronize bracket adds to a short timed region: n launch pairs (hsvfilter, colorlut; 8 x 4K) between
synchronize() calls - host clock across the bracket against the device's own span between an event in front of the first launch
and one behind the last; with the device idle before the bracket (as synchronize leaves it) and with one untimed launch queued
in front. Run on the GPU box: python tools/bracket_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

W, H, N = bench.W, bench.H, 8
dev = torch.device("cuda:0")
ctx = mi355fx.Context(0)
stream = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
pool_ev = bench.EventPool(torch, 256)
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
settings = synth.HSV_SETTINGS["hue90"]
pool = bench.SourcePool(torch, synth, dev, N, "smooth")
srcs = [pool.new(k) for k in range(96)]
dsts = [torch.empty_like(srcs[0]) for _ in range(4)]
pitch = W * H * 4
k = 0


def pairs(n):
    global k
    for _ in range(n):
        s = srcs[k % len(srcs)]
        ctx.hsvfilter_frames_device(s.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
        ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, dsts[k % 4].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
        k += 1


t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    pairs(64); torch.cuda.synchronize()
for n in (80, 300):
    for rep in range(4):
        pairs(150)                      # the rewarm of bench.py
        torch.cuda.synchronize()
        e0, e1 = pool_ev.take(), pool_ev.take()
        ta = time.perf_counter()
        e0.record()
        pairs(n)
        tb = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        dev_ms = e0.elapsed_time(e1)
        print("%3d pairs: host bracket %.3f ms (%.1f us per pair); device span %.3f ms (%.1f us per pair); host enqueue %.2f ms; bracket - span %.3f ms" %
              (n, (tc - ta) * 1e3, (tc - ta) / n * 1e6, dev_ms, dev_ms / n * 1e3, (tb - ta) * 1e3, (tc - ta) * 1e3 - dev_ms))
