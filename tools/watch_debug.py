"""Development tool: the content watch of the interpolating colorlut path inside the bench's chain, launches enqueued without
host synchronisation (the host runs far ahead of the device, so snapshots arrive many launches late)."""
import sys, os, numpy as np, torch, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'gst-plugins-rs_amd'))
import mi355fx, bench
from mi355fx import synth
from mi355fx.cube import parse_cube
dev = torch.device('cuda', 0)
W, H = 3840, 2160
lut = parse_cube(synth.cube_text_3d(33))
ctx = mi355fx.Context(0)
st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    ctx.set_stream(st.cuda_stream)
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 6)
    pool = bench.SourcePool(torch, synth, dev, 8, 'smooth')
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    srcs = [pool.new(k) for k in range(n)]
    dst = torch.empty_like(srcs[0])
    hs = synth.HSV_SETTINGS['hue90']
    torch.cuda.synchronize()
    names = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(n):
        ctx.hsvfilter_frames_device(srcs[k].data_ptr(), 8, W * H * 4, W, H, W * 4, "RGBA", hs)
        ctx.colorlut_frames_device(srcs[k].data_ptr(), W * H * 4, W * 4, dst.data_ptr(), W * H * 4, W * 4, 8, W, H, "RGBA")
        names.append(ctx.colorlut_kernel_name().replace('colorlut3d_', ''))
    e1.record()
    torch.cuda.synchronize()
    runs = []
    for nm in names:
        if runs and runs[-1][0] == nm: runs[-1][1] += 1
        else: runs.append([nm, 1])
    print("ms per step %.4f" % (e0.elapsed_time(e1) / n))
    print(runs)
    print(ctx.colorlut_brick_stats())
