"""Profiling driver: N launches of one colorlut kernel variant on 8x4K natural-like (or noise) frames, nothing else.
   python3 tools/run_colorlut_once.py <variant> [launches] [amp|-1] [sets]      (under rocprofv3: put python3 first)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 7
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 20
amp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sets = int(sys.argv[4]) if len(sys.argv) > 4 else 32
ctx = mi355fx.Context(0)
lut = parse_cube(synth.cube_text_3d(33))
ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
if os.environ.get("ORDER"):   # MI355_FLAG_WINDOW_ORDER of the LDS-cached table kernel (variant 8)
    ctx.set_flag(mi355fx.FLAG_WINDOW_ORDER, int(os.environ["ORDER"]))
if amp < 0:
    frames = np.stack([synth.noise_frame(W, H, seed=11 + i) for i in range(N)]).reshape(-1)
else:
    f = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
    if os.environ.get("ROT"):   # bench.py's batches: one base frame, rotated along the row by 97 i pixels
        f = np.stack([np.roll(f[0], 97 * i, axis=1) for i in range(N)])
    if amp:
        f[..., :3] += np.random.default_rng(2).integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
    frames = np.clip(f, 0, 255).astype(np.uint8).reshape(-1)
d_src, d_dst = ctx.alloc(frames.size), ctx.alloc(frames.size)
ctx.h2d(d_src, frames)
if os.environ.get("PREHSV"):  # the chain's colorlut input
    ctx.hsvfilter_frames_device(d_src, N, H * W * 4, W, H, W * 4, "RGBA", synth.HSV_SETTINGS["hue90"])
    ctx.synchronize()
ms = ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", launches)
print("variant %d: %.4f ms per launch, kernel %s" % (variant, ms, ctx.colorlut_kernel_name()))
ctx.free(d_src); ctx.free(d_dst); ctx.close()
