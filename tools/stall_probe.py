"""Where does the host block while it enqueues the chain? Per-iteration host time of 300 steps x 4 (hsvfilter, colorlut) pairs,
auto kernel choice, optional timing events every 4th step as bench.py records them; prints the iterations that took more than
1 ms on the host and the device-side total. Run on the GPU box: python tools/stall_probe.py [events=1] [gc=1] [batches=64]"""
import gc, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

use_events = int(sys.argv[1]) if len(sys.argv) > 1 else 1
use_gc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n_batches = int(sys.argv[3]) if len(sys.argv) > 3 else 64
if not use_gc:
    gc.disable()
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
srcs = [pool.new(k) for k in range(n_batches)]
dsts = [torch.empty_like(srcs[0]) for _ in range(4)]
pitch = W * H * 4


def pair(k):
    s = srcs[k % n_batches]
    ctx.hsvfilter_frames_device(s.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
    ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, dsts[k % 4].data_ptr(), pitch, W * 4, N, W, H, "RGBA")


t0 = time.perf_counter()
k = 0
while time.perf_counter() - t0 < 0.4:
    for _ in range(64):
        pair(k); k += 1
    torch.cuda.synchronize()
for rep in range(3):
    times, evs = [], []
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for step in range(300):
        h0 = time.perf_counter()
        if use_events and step % 4 == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for _ in range(4):
            pair(k); k += 1
        if use_events and step % 4 == 0:
            e1.record(); evs.append((e0, e1))
        times.append(time.perf_counter() - h0)
    tb = time.perf_counter()
    torch.cuda.synchronize()
    tc = time.perf_counter()
    slow = [(i, t * 1e3) for i, t in enumerate(times) if t > 1e-3]
    print("events %d gc %d: host enqueue %.1f ms, device done after %.1f ms (%.0f frames/s); host iterations > 1 ms: %s" %
          (use_events, use_gc, (tb - ta) * 1e3, (tc - ta) * 1e3, 1200 * N / (tc - ta), ["step %d: %.1f ms" % s for s in slow][:12]))
