"""A/B timing of the colorlut kernel kinds and variants (MI355_FLAG_LUT_VARIANT) on the headline batch (8 x 4K RGBA, 33^3).
Run on the GPU box: python tools/lut_variants.py"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, int(os.environ.get("N", "8"))
NAMES = {6: "interpolating (5 regs/px, 1024x3)", 0: "auto", 2: "lean state (2 regs/px, 1024x8)", 4: "full table, linear index", 5: "full table, Morton index"}


def main():
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    for content in os.environ.get("CONTENT", "smooth,noise").split(","):
        mk = synth.smooth_frame if content == "smooth" else synth.noise_frame
        frames = np.stack([mk(W, H, seed=7 + i) for i in range(N)]).reshape(-1)
        d_src, d_dst = ctx.alloc(frames.nbytes), ctx.alloc(frames.nbytes)
        ctx.h2d(d_src, frames)
        ref = None
        for rep in range(2):
            for v in [int(x) for x in os.environ.get("VARIANTS", "6,2,4,5,0").split(",")]:
                ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
                ms = min(ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 20) for _ in range(3))
                out = np.zeros_like(frames)
                ctx.d2h(out, d_dst)
                crc = zlib.crc32(out.tobytes())
                ref = crc if ref is None else ref
                print("%-7s variant %d (%-30s) %.4f ms  %.0f GB/s  crc %s" % (content, v, NAMES[v], ms, N * W * H * 8 / ms / 1e6, "ok" if crc == ref else "MISMATCH"), ctx.colorlut_kernel_choice() if v == 0 else "", flush=True)
        ctx.free(d_src); ctx.free(d_dst)
    ctx.close()


if __name__ == "__main__":
    main()
