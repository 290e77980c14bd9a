"""A/B of the memoised-table kernels on 8 x 4K RGBA (33^3): Morton table through the tiled gather kernel (variant 5), window
table through the tiled gather kernel (9), window table through the LDS-cached kernel (8), brick kernel (6); on the
natural-like frame + uniform noise of +-amp, pristine and after hsvfilter (the chain's input); plus the fused entry point.
Prints ms per launch and the LDS cache's statistics. Run on the GPU box: python tools/window_probe.py [amps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8


def main():
    amps = [int(a) for a in sys.argv[1:]] or [0, 4, 8, 16]
    # "8:0" / "8:1": variant 8 (colorlut_window_kernel) with MI355_FLAG_WINDOW_ORDER 0 (contiguous shares) / 1 (aligned fronts)
    specs = os.environ.get("VARIANTS", "5,8:0,8:1,6").split(",")
    variants = list(range(len(specs)))
    vk = [(int(x.split(":")[0]), int(x.split(":")[1]) if ":" in x else 1) for x in specs]
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    rng = np.random.default_rng(2)
    base = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
    d_src, d_dst = ctx.alloc(base.size), ctx.alloc(base.size)
    st = synth.HSV_SETTINGS["hue90"]
    print("%-10s %-5s " % ("input", "amp") + " ".join("v%-9s" % specs[v] for v in variants) + " window: past-cache %  installs/step")
    for amp in amps:
        f = base.copy()
        if amp:
            f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
        frames = np.clip(f, 0, 255).astype(np.uint8).reshape(-1)
        for post in (False, True):
            ctx.h2d(d_src, frames)
            if post:
                ctx.hsvfilter_frames_device(d_src, N, H * W * 4, W, H, W * 4, "RGBA", st)
                ctx.synchronize()
            res = {}
            stats = ""
            for v in variants:
                ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
                ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, vk[v][0])
                ctx.set_flag(mi355fx.FLAG_WINDOW_ORDER, vk[v][1])
                ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 3)
                res[v] = min(ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 20) for _ in range(3))
                if vk[v][0] == 8:
                    ctx.set_flag(mi355fx.FLAG_WINDOW_STATS, 1)
                    ctx.colorlut_window_stats(reset=True)
                    ctx.time_colorlut_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, "RGBA", 2)
                    ctx.set_flag(mi355fx.FLAG_WINDOW_STATS, 0)
                    px, past, inst = ctx.colorlut_window_stats()
                    stats += " [%s] %.2f %%  %.1f" % (specs[v], 100.0 * past / max(px, 1), inst / max(px / 8192.0, 1))
            print("%-10s %-5d " % ("post-hsv" if post else "pristine", amp) + " ".join("%-10.4f" % res[v] for v in variants) + " " + stats, flush=True)
    # the fused entry point on pristine frames (one launch for the chain)
    ctx.h2d(d_src, np.clip(base, 0, 255).astype(np.uint8).reshape(-1))
    for v in variants:
        ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, vk[v][0])
        ctx.set_flag(mi355fx.FLAG_WINDOW_ORDER, vk[v][1])
        for _ in range(3):
            ctx.hsv_colorlut_frames_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, st)
        ctx.synchronize()
        import time
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                ctx.hsv_colorlut_frames_device(d_src, H * W * 4, W * 4, d_dst, H * W * 4, W * 4, N, W, H, st)
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        print("fused pristine amp 0  variant %s: %.4f ms per launch (host clock, 20 launches)  kernel %s" % (specs[v], best, ctx.colorlut_kernel_name()), flush=True)
    ctx.free(d_src); ctx.free(d_dst); ctx.close()


if __name__ == "__main__":
    main()
