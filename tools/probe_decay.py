"""How many launches does the LDS-cached table kernel need after a stretch of gather-kernel launches before it runs at its own
speed? (The run-time choice probes the kind not in use with TWO launches and measures the second.) Fused entry point, 8 x 4K from
HBM, fresh batch per launch. Run on the GPU box: python tools/probe_decay.py"""
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
evp = bench.EventPool(torch, 512)
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
settings = synth.HSV_SETTINGS["hue90"]
pool = bench.SourcePool(torch, synth, dev, N, "smooth")
srcs = [pool.new(k) for k in range(48)]
dsts = [torch.empty_like(srcs[0]) for _ in range(4)]
pitch = W * H * 4
k = 0
fused = int(os.environ.get("FUSED", "1"))


def launch(timed=None):
    global k
    s = srcs[k % len(srcs)]; d = dsts[k % 4]; k += 1
    if timed is not None:
        e0, e1 = evp.take(), evp.take(); e0.record()
    if fused:
        ctx.hsv_colorlut_frames_device(s.data_ptr(), pitch, W * 4, d.data_ptr(), pitch, W * 4, N, W, H, settings)
    else:
        ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, d.data_ptr(), pitch, W * 4, N, W, H, "RGBA")
    if timed is not None:
        e1.record(); timed.append((e0, e1))


for v in (5, 8):   # tables built, kernels loaded
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
    for _ in range(24):
        launch()
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
while time.perf_counter() - t0 < 0.4:
    for _ in range(32):
        launch()
    torch.cuda.synchronize()
def group(n_launch, reps):
    out = []
    for _ in range(reps):
        e0, e1 = evp.take(), evp.take(); e0.record()
        for _ in range(n_launch):
            launch()
        e1.record(); out.append((e0, e1))
    return out


for gsz in (1, 2, 4):
    res = {}
    for v in (5, 8, 5, 8):
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
        for _ in range(8):
            launch()
        res.setdefault(v, []).extend(group(gsz, 12))
    torch.cuda.synchronize()
    med = lambda l: sorted(a.elapsed_time(b) / gsz for a, b in l)[len(l) // 2]
    print("brackets around %d consecutive launches: gather %.4f ms per launch, window %.4f (%.1f %% less)" % (gsz, med(res[5]), med(res[8]), 100 * (1 - med(res[8]) / med(res[5]))))
for rep in range(1):
    ev = []
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    for _ in range(64):
        launch()
    g = []
    for _ in range(4):
        launch(g)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 8)
    for _ in range(10):
        launch(ev)
    names = ctx.colorlut_kernel_name()
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)
    h = []
    for _ in range(4):
        launch(h)
    torch.cuda.synchronize()
    print("gather (last 4 of 68): %s | window launches 1..10 after them: %s | gather again 1..4: %s   [%s]" %
          (" ".join("%.4f" % a.elapsed_time(b) for a, b in g), " ".join("%.4f" % a.elapsed_time(b) for a, b in ev), " ".join("%.4f" % a.elapsed_time(b) for a, b in h), names))
