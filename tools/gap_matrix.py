"""Gap between consecutive launches on one stream by kernel pair: runs of hsvfilter only, colorlut (gather kernel) only, and the
alternating chain, 8 x 4K, kernels pinned. Run under rocprofv3 --kernel-trace and feed the trace to tools/launch_gaps.py:
  rocprofv3 --kernel-trace --output-format csv -d /tmp/gm -o t -- python3 tools/gap_matrix.py ; python3 tools/launch_gaps.py /tmp/gm 100000"""
import os, sys
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
srcs = [pool.new(k) for k in range(16)]
dsts = [torch.empty_like(srcs[0]) for _ in range(2)]
pitch = W * H * 4
hsv = lambda k: ctx.hsvfilter_frames_device(srcs[k % 16].data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
lutk = lambda k: ctx.colorlut_frames_device(srcs[k % 16].data_ptr(), pitch, W * 4, dsts[k % 2].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
for k in range(64):
    hsv(k); lutk(k)
torch.cuda.synchronize()
for rep in range(3):
    for k in range(200):
        hsv(k)
    for k in range(200):
        lutk(k)
    for k in range(200):
        hsv(k); lutk(k)
    torch.cuda.synchronize()
