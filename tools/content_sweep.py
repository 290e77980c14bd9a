"""colorlut kernels against content with less and less colour locality: the natural-like frame with uniform noise of
+-amp added per channel. Per 8x4K launch: interpolating kernel (variant 6), table kernel (variant 5), auto (variant 0)
and what auto ended up using. Run on the GPU box: python tools/content_sweep.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8


def main():
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    rng = np.random.default_rng(2)
    base = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
    d_src, d_dst = ctx.alloc(base.size), ctx.alloc(base.size)
    print("%-8s %-12s %-12s %-12s %s" % ("amp", "interp ms", "table ms", "auto ms", "auto uses"))
    for amp in [int(a) for a in os.environ.get("AMPS", "0,2,4,8,16,24,32,48,64,128").split(",")]:
        f = base.copy()
        if amp:
            f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
        frames = np.clip(f, 0, 255).astype(np.uint8).reshape(-1)
        ctx.h2d(d_src, frames)
        res = {}
        for v in (6, 5, 0):
            ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)  # fresh auto state
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
            res[v] = min(ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 40) for _ in range(3))
        print("%-8d %-12.4f %-12.4f %-12.4f %s" % (amp, res[6], res[5], res[0], "table" if ctx.colorlut_kernel_choice()[0] else "interpolating"), flush=True)
    ctx.free(d_src); ctx.free(d_dst); ctx.close()


if __name__ == "__main__":
    main()
