"""What the nested run-time choice (gather kernel / LDS-cached kernel for the memoised table) measures over time, fused entry
point, 8 x 4K from HBM, fresh batch per launch, no brackets of its own. Run on the GPU box: python tools/inner_choice_probe.py"""
import os, sys, time
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
srcs = [pool.new(k) for k in range(48)]
dsts = [torch.empty_like(srcs[0]) for _ in range(4)]
pitch = W * H * 4
fused = int(os.environ.get("FUSED", "1"))
k = 0
names = {}
for block in range(24):
    t0 = time.perf_counter()
    for _ in range(50):
        s = srcs[k % 48]; d = dsts[k % 4]; k += 1
        if fused:
            ctx.hsv_colorlut_frames_device(s.data_ptr(), pitch, W * 4, d.data_ptr(), pitch, W * 4, N, W, H, settings)
        else:
            ctx.colorlut_frames_device(s.data_ptr(), pitch, W * 4, d.data_ptr(), pitch, W * 4, N, W, H, "RGBA")
        nm = ctx.colorlut_kernel_name(); names[nm] = names.get(nm, 0) + 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50 * 1e3
    outer = ctx.colorlut_kernel_choice(fused=1 if fused else 0)
    inner = ctx.colorlut_kernel_choice(fused=11 if fused else 10)
    mpx = N * W * H / 1e6
    print("launches %4d: %.4f ms per launch | outer table=%s compute %.4f table %.4f | inner window=%s gather %.4f window %.4f | %s" %
          (k, dt, outer[0], outer[1] * mpx, outer[2] * mpx, inner[0], inner[1] * mpx, inner[2] * mpx, names))
    names = {}
