"""Do the memoised-table gather kernel (texture-address bound) and the interpolating brick kernel (VALU bound) ADD UP when they run on
the same CUs at the same time? 8 x 4K RGBA (33^3) over content noise: the table kernel alone, the interpolating kernel alone, and the
batch split by frames - k frames through the table kernel on one context's stream, 8 - k through the interpolating kernel on another's,
enqueued back to back, nobody waits in between. Wall clock per batch over 20 batches. Run on the GPU box: python tools/split_probe.py [amps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N = 3840, 2160, 8
FB = W * H * 4
B_FIRST = os.environ.get("B_FIRST", "0") == "1"   # enqueue the interpolating kernel's launch before the table kernel's


def main():
    amps = [int(a) for a in sys.argv[1:]] or [0, 4, 8]
    a, b = mi355fx.Context(0), mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    for c, variant in ((a, 5), (b, int(os.environ.get("COMPUTE_VARIANT", "6")))):
        c.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        c.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
    rng = np.random.default_rng(2)
    base = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
    d_src, d_dst = a.alloc(N * FB), a.alloc(N * FB)
    st = synth.HSV_SETTINGS["hue90"]

    def run(k, iters=20):
        """k frames through the table kernel (context a), N - k through the interpolating kernel (context b)"""
        def once():
            if k < N and B_FIRST:
                b.colorlut_frames_device(d_src + k * FB, FB, W * 4, d_dst + k * FB, FB, W * 4, N - k, W, H, "RGBA")
            if k:
                a.colorlut_frames_device(d_src, FB, W * 4, d_dst, FB, W * 4, k, W, H, "RGBA")
            if k < N and not B_FIRST:
                b.colorlut_frames_device(d_src + k * FB, FB, W * 4, d_dst + k * FB, FB, W * 4, N - k, W, H, "RGBA")
        for _ in range(3):
            once()
        a.synchronize(); b.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(iters):
                once()
            a.synchronize(); b.synchronize()
            best = min(best, (time.perf_counter() - t0) / iters * 1e3)
        return best

    for amp in amps:
        f = base.copy()
        if amp:
            f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
        frames = np.clip(f, 0, 255).astype(np.uint8).reshape(-1)
        a.h2d(d_src, frames)
        a.hsvfilter_frames_device(d_src, N, FB, W, H, W * 4, "RGBA", st)
        a.synchronize()
        t_tab, t_cmp = run(N), run(0)
        row = "  ".join("%d+%d: %.4f" % (k, N - k, run(k)) for k in (3, 4, 5, 6))
        print("amp %-2d  table alone %.4f (%s)  interpolating alone %.4f (%s)  harmonic %.4f | split table+interpolating frames  %s" %
              (amp, t_tab, a.colorlut_kernel_name(), t_cmp, b.colorlut_kernel_name(), t_tab * t_cmp / (t_tab + t_cmp), row), flush=True)
    a.free(d_src); a.free(d_dst); a.close(); b.close()


if __name__ == "__main__":
    main()
