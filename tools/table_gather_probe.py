"""What bounds colorlut_table_kernel: time per 8x4K for frames whose colours come from palettes of different sizes
(1 colour: every lane gathers the same address; 64 / 4096 random colours: distinct lines that stay in L1 / L2;
2^24: every gather misses). Run on the GPU box: python tools/table_gather_probe.py"""
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
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    rng = np.random.default_rng(1)
    n_px = W * H * N
    d_src, d_dst = ctx.alloc(n_px * 4), ctx.alloc(n_px * 4)
    for name, pal in (("1 colour", 1), ("16 colours", 16), ("64 colours", 64), ("1024 colours", 1024), ("4096 colours", 4096),
                      ("65536 colours", 65536), ("1M colours", 1 << 20), ("2^24 colours", 1 << 24),
                      ("runs of 4 equal pixels, 4096 colours", -4096), ("runs of 64 equal pixels, 4096 colours", -4096 * 64)):
        if pal > 0:
            palette = rng.integers(0, 1 << 24, size=pal, dtype=np.uint32)
            px = palette[rng.integers(0, pal, size=n_px)]
        else:
            run = 4 if pal == -4096 else 64
            palette = rng.integers(0, 1 << 24, size=4096, dtype=np.uint32)
            px = np.repeat(palette[rng.integers(0, 4096, size=n_px // run)], run)
        frames = (px | np.uint32(0xFF000000)).view(np.uint8)
        ctx.h2d(d_src, frames)
        for v in (5, 4, 6):
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
            ms = min(ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 20) for _ in range(3))
            print("%-40s variant %d  %.4f ms  %.2f Gpx/s" % (name, v, ms, n_px / ms / 1e6), flush=True)
    ctx.free(d_src); ctx.free(d_dst); ctx.close()


if __name__ == "__main__":
    main()
