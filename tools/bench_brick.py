"""Interpolating colorlut kernels on 8x4K: brick-cache kernel (variant 7) against the three-pass whole-plane kernel
(variant 3) and the memoised-table kernel (5), on the natural-like frame with uniform noise of +-amp added per channel,
and on pure noise. Prints ms per launch, fraction of the 8 TB/s HBM peak (8 B/pixel algorithmic) and the brick kernel's
careful-path share. Run on the GPU box: python tools/bench_brick.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, int(os.environ.get("N", "8"))


def main():
    ctx = mi355fx.Context(0)
    size = int(os.environ.get("LUT_SIZE", "33"))
    lut = parse_cube(synth.cube_text_3d(size))
    rng = np.random.default_rng(2)
    base = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
    d_src, d_dst = ctx.alloc(base.size), ctx.alloc(base.size)
    variants = [int(v) for v in os.environ.get("VARIANTS", "7,3,5").split(",")]
    tprs = [int(v) for v in os.environ.get("TPR", "0").split(",")]
    setss = [int(v) for v in os.environ.get("SETS", "32,64").split(",")]
    algo = 8.0 * W * H * N
    print("%-8s %s" % ("amp", "  ".join("v%d[tpr %d sets %d] ms (frac)" % (v, t, ns) for v in variants for t in (tprs if v == 7 else [0]) for ns in (setss if v == 7 else [0]))))
    for amp in [int(a) for a in os.environ.get("AMPS", "0,2,4,8,16,32,64,-1").split(",")]:
        if amp < 0:
            frames = np.stack([synth.noise_frame(W, H, seed=11 + i) for i in range(N)]).reshape(-1)
        else:
            f = base.copy()
            if amp:
                f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
            frames = np.clip(f, 0, 255).astype(np.uint8).reshape(-1)
        ctx.h2d(d_src, frames)
        if os.environ.get("PREHSV"):  # the chain's colorlut input: the frames after hsvfilter(hue-shift=90)
            ctx.hsvfilter_frames_device(d_src, N, H * W * 4, W, H, W * 4, "RGBA", synth.HSV_SETTINGS["hue90"])
            ctx.synchronize()
        cells = []
        for v in variants:
            for t, ns in [(t, ns) for t in (tprs if v == 7 else [0]) for ns in (setss if v == 7 else [0])]:
                ctx.set_flag(mi355fx.FLAG_BRICK_FOLD_AXIS, int(os.environ.get("FOLD", "2")))
                ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
                ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
                ctx.set_flag(mi355fx.FLAG_BRICK_TILES_PER_RUN, t)
                ctx.set_flag(mi355fx.FLAG_BRICK_SETS, ns)
                ctx.set_flag(mi355fx.FLAG_BRICK_PRIO, int(os.environ.get("PRIO", "3")))
                if os.environ.get("FUSED"):  # the fused hsvfilter -> colorlut launch (compute kernels; the composed table is not built here)
                    run = lambda it: ctx.time_hsv_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, synth.HSV_SETTINGS["hue90"], it)
                else:
                    run = lambda it: ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", it)
                mi355fx.warm_clocks(lambda: run(1), ctx.synchronize, 0.2)
                ctx.colorlut_brick_stats(reset=True)
                ms = min(run(20) for _ in range(3))
                steps, slow, _, _ = ctx.colorlut_brick_stats(reset=True)
                extra = " miss %.3f slow %.3f" % (steps / 60.0 / (W * H * N / 256), slow / 60.0 / (W * H * N / 256)) if v == 7 else ""
                if v in (0, 6):
                    extra = " [%s, watch level %d]" % (ctx.colorlut_kernel_name().replace("colorlut", ""), ctx.colorlut_brick_stats()[3])
                cells.append("%.4f (%.3f)%s" % (ms, algo / (ms * 1e-3) / 8e12, extra))
        print("%-8s %s" % ("noise" if amp < 0 else amp, "  ".join(cells)), flush=True)
    ctx.free(d_src); ctx.free(d_dst); ctx.close()


if __name__ == "__main__":
    main()
