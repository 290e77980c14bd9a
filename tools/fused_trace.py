"""Per-launch times of the fused hsvfilter->colorlut entry point under the auto kernel choice (diagnostic).
Run on the GPU box: python tools/fused_trace.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

W, H = 3840, 2160
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
every = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda", 0)
ctx = mi355fx.Context(0)
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx.set_stream(stream.cuda_stream)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    srcs = bench.make_batches(torch, synth, dev, batch, 4, "smooth")
    dsts = [torch.empty_like(s) for s in srcs]
    st = synth.HSV_SETTINGS["hue90"]
    pitch = W * H * 4
    evs = []
    for k in range(60):
        s_, d_ = srcs[k % 4], dsts[k % 4]
        if k % every == 0:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        ctx.hsv_colorlut_frames_device(s_.data_ptr(), pitch, W * 4, d_.data_ptr(), pitch, W * 4, batch, W, H, st)
        if k % every == 0:
            e1.record()
            evs.append((k, e0, e1))
    torch.cuda.synchronize()
    print(" ".join("%d:%.3f" % (k, a.elapsed_time(b)) for k, a, b in evs))
    print(ctx.colorlut_kernel_choice(fused=True))
ctx.close()
