"""A/B timing of the fused hsvfilter+colorlut tilings (MI355_FLAG_FUSED_VARIANT) on the headline batch.
Run on the GPU box:  python tools/fused_variants.py"""
import os, sys, zlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8
NAMES = {0: "inline hsv, 1024x3", 1: "pipelined, 1024x2"}


def main():
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    st = synth.HSV_SETTINGS["hue90"]
    for content in ("smooth", "noise"):
        mk = synth.smooth_frame if content == "smooth" else synth.noise_frame
        frames = np.stack([mk(W, H, seed=7 + i) for i in range(N)]).reshape(-1)
        d_src, d_dst = ctx.alloc(frames.nbytes), ctx.alloc(frames.nbytes)
        ctx.h2d(d_src, frames)
        ref = None
        for v in (0, 1):
            ctx.set_flag(mi355fx.FLAG_FUSED_VARIANT, v)
            ms = min(ctx.time_hsv_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, st, 20) for _ in range(3))
            out = np.zeros_like(frames)
            ctx.d2h(out, d_dst)
            crc = zlib.crc32(out.tobytes())
            ref = crc if ref is None else ref
            print("%-7s variant %d (%-18s) %.4f ms  %.0f fps  crc %s" % (content, v, NAMES[v], ms, N / ms * 1e3, "ok" if crc == ref else "MISMATCH"), flush=True)
        ctx.free(d_src); ctx.free(d_dst)
    ctx.close()


if __name__ == "__main__":
    main()
