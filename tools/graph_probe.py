"""Do hipGraphs shrink the gaps between the dependent launches of the chain? Eight (hsvfilter, colorlut) pairs over eight batches,
kernels pinned (table kernel, no run-time choice: nothing but launches on the stream), issued as stream launches and as ONE
captured graph replayed. Run on the GPU box: python tools/graph_probe.py"""
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
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
settings = synth.HSV_SETTINGS["hue90"]
pool = bench.SourcePool(torch, synth, dev, N, "smooth")
srcs = [pool.new(k) for k in range(8)]
dsts = [torch.empty_like(srcs[0]) for _ in range(2)]
pitch = W * H * 4


def pairs(n):
    for k in range(n):
        s = srcs[k % 8]
        ctx.hsvfilter_frames_device(s.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
        ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, dsts[k % 2].data_ptr(), pitch, W * 4, N, W, H, "RGBA")


pairs(16); torch.cuda.synchronize()   # table built, kernels loaded
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.4:
    pairs(64); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter(); pairs(400); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("stream launches: %.1f us per pair (%.0f frames/s)" % ((t1 - t0) / 400 * 1e6, 400 * N / (t1 - t0)))
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        pairs(8)
    torch.cuda.synchronize()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print("graph of 8 pairs replayed: %.1f us per pair (%.0f frames/s)" % ((t1 - t0) / 400 * 1e6, 400 * N / (t1 - t0)))
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", repr(e)[:300])
