"""How far ahead of the device is the host? Enqueue time per (hsvfilter, colorlut) launch pair from the interpreter against the
device time per pair, 8 x 4K per launch, auto kernel choice - if the first approaches the second the bench measures the caller.
Run on the GPU box: python tools/host_enqueue_probe.py"""
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


pairs(400); torch.cuda.synchronize()
for rep in range(4):
    n = 400
    t0 = time.perf_counter(); pairs(n); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("enqueue %.1f us per pair (host), %.1f us per pair until the device is done; kernel %s" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6, ctx.colorlut_kernel_name()))
# the same from one native loop per round of 8 one-frame streams is mi355_issue_streams_round; here: cost of the two ctypes calls alone
t0 = time.perf_counter()
for _ in range(2000):
    ctx.colorlut_kernel_name()
t1 = time.perf_counter()
print("a trivial ctypes call: %.2f us" % ((t1 - t0) / 2000 * 1e6))
print("host cores: %d, load average %s" % (os.cpu_count(), os.getloadavg()))
