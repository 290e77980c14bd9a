"""Effect of de-synchronising the CUs of the colorlut 3D LDS kernel (MI355_FLAG_LUT_STAGGER) on 8 x 4K RGBA, 33^3."""
import os, sys, zlib
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
    for content in ("smooth", "noise"):
        mk = synth.smooth_frame if content == "smooth" else synth.noise_frame
        frames = [np.stack([mk(W, H, seed=7 + i + 100 * r) for i in range(N)]).reshape(-1) for r in range(2)]
        d_src = [ctx.alloc(f.nbytes) for f in frames]
        d_dst = [ctx.alloc(f.nbytes) for f in frames]
        for d, f in zip(d_src, frames):
            ctx.h2d(d, f)
        ref = None
        for rep in range(2):
            for stg in (0, 16, 32, 64, 96, 128, 192):
                ctx.set_flag(mi355fx.FLAG_LUT_STAGGER, stg)
                best = 1e9
                for _ in range(3):   # alternate two 265 MB batches so the working set exceeds the Infinity Cache
                    ms = 0.5 * (ctx.time_colorlut_device(d_src[0], H * W * 4, W * 4, d_dst[0], H * W * 4, W * 4, N, W, H, "RGBA", 10) +
                                ctx.time_colorlut_device(d_src[1], H * W * 4, W * 4, d_dst[1], H * W * 4, W * 4, N, W, H, "RGBA", 10))
                    best = min(best, ms)
                out = np.zeros_like(frames[0]); ctx.d2h(out, d_dst[0])
                crc = zlib.crc32(out.tobytes()); ref = crc if ref is None else ref
                print("%-7s stagger %4d x256 ticks  %.4f ms  %.0f GB/s  crc %s" % (content, stg, best, N * W * H * 8 / best / 1e6, "ok" if crc == ref else "MISMATCH"), flush=True)
        for d in d_src + d_dst:
            ctx.free(d)
    ctx.close()


if __name__ == "__main__":
    main()
