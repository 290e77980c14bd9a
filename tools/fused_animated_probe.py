"""hsvfilter ! colorlut fused (mi355_hsv_colorlut_frames_device) on 8 x 4K with a hue shift that changes on EVERY call - what a
controller animating `hue-shift` produces. ms per call under auto against the arithmetic kernel pinned (MI355_FLAG_LUT_VARIANT 6), and
with settings that stay. (Round 6 ran it on a probe build in which a launch of 40 Mpixel or more built the composed table for its own
settings instead of waiting for them to settle: the "auto" lines of profiles/r06_fused_animated_probe.txt. Slower; not kept.) Run on the GPU box: python tools/fused_animated_probe.py"""
import os, sys, time
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
    src = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(-1)
    d_s, d_o = ctx.alloc(src.nbytes), ctx.alloc(src.nbytes)
    ctx.h2d(d_s, src)
    pitch = W * H * 4

    def run(variant, animated, calls=60):
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
        def call(k):
            st = (10.0 + (0.5 * k if animated else 0.0), 1.0, 0.0, 1.0, 0.0)
            ctx.hsv_colorlut_frames_device(d_s, pitch, W * 4, d_o, pitch, W * 4, N, W, H, st)
        for k in range(40):
            call(k)
        ctx.synchronize()
        t0 = time.perf_counter()
        for k in range(calls):
            call(100 + k)
        ctx.synchronize()
        return (time.perf_counter() - t0) / calls * 1e3, ctx.colorlut_kernel_name()

    for rep in range(2):
        for label, variant, animated in (("settings stay, auto", 0, False), ("hue shift changes every call, auto", 0, True),
                                         ("hue shift changes every call, arithmetic kernel pinned (rounds 2-5)", 6, True)):
            ms, name = run(variant, animated)
            print("%-75s %.4f ms per 8 x 4K call  (%s)" % (label, ms, name), flush=True)
    ctx.free(d_s); ctx.free(d_o); ctx.close()


if __name__ == "__main__":
    main()
