"""colorlut RGBA64 (16 B/pixel algorithmic) on 8 x 4K, warm clocks: 33^3 / 17^3 / 65^3 / 1D. Run on the GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8
ctx = mi355fx.Context(0)
rng = np.random.default_rng(0)
for content in ("smooth", "noise"):
    if content == "smooth":
        f8 = np.stack([synth.smooth_frame(W, H, seed=i) for i in range(N)]).reshape(-1).astype(np.uint16)
        frames = (f8 * 257 + rng.integers(0, 64, f8.shape, dtype=np.uint16)).astype(np.uint16)
    else:
        frames = rng.integers(0, 65536, N * W * H * 4, dtype=np.uint16)
    nb = frames.nbytes
    d_s, d_o = ctx.alloc(nb), ctx.alloc(nb)
    ctx.h2d(d_s, frames.view(np.uint8))
    for name, text in (("33^3", synth.cube_text_3d(33)), ("17^3", synth.cube_text_3d(17)), ("65^3", synth.cube_text_3d(65)), ("1D 1024", synth.cube_text_1d(1024))):
        lut = parse_cube(text)
        ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        mi355fx.warm_clocks(lambda: ctx.colorlut_frames_device(d_s, W * H * 8, W * 8, d_o, W * H * 8, W * 8, N, W, H, "RGBA64_LE"), ctx.synchronize)
        ms = min(ctx.time_colorlut_device(d_s, W * H * 8, W * 8, d_o, W * H * 8, W * 8, N, W, H, "RGBA64_LE", 30) for _ in range(3))
        print("%-7s RGBA64_LE %-8s %.4f ms  %.0f GB/s (%.1f %% of 8 TB/s)  [%s]" % (content, name, ms, 2 * nb / ms / 1e6, 2 * nb / ms / 8e7, ctx.colorlut_kernel_name()), flush=True)
        if lut.is3d:  # the three-pass 16-bit kernel pinned, for comparison
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 3)
            ms = min(ctx.time_colorlut_device(d_s, W * H * 8, W * 8, d_o, W * H * 8, W * 8, N, W, H, "RGBA64_LE", 30) for _ in range(2))
            print("%-7s RGBA64_LE %-8s %.4f ms  (%.1f %%)  [%s pinned]" % (content, name, ms, 2 * nb / ms / 8e7, ctx.colorlut_kernel_name()), flush=True)
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 0)
    ctx.free(d_s); ctx.free(d_o)
ctx.close()
