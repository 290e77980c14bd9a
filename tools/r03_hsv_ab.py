"""Round-3 A/B of the hsvfilter kernel on the GPU box: blocks per CU x LDS table variant (MI355_FLAG_HSV_TAB),
8 x 4K RGBA in place, hue-shift 90; plus GENERIC / wide-shift / 3-byte / padded-row timings (tools, not product)."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "gst-plugins-rs_amd"))
import numpy as np, mi355fx
from mi355fx import synth
W, H, B = 3840, 2160, 8
ctx = mi355fx.Context(0)
one = synth.smooth_frame(W, H)
fr = np.stack([one] * B)
bufs = [ctx.alloc(fr.nbytes) for _ in range(4)]
def refill():
    for d in bufs: ctx.h2d(d, fr)
refill()
st = synth.HSV_SETTINGS["hue90"]
def t(d, settings, fmt="RGBA", iters=40, pitch=W*H*4, w=W, h=H, stride=W*4, n=B):
    ctx.time_hsvfilter_device(d, n, pitch, w, h, stride, fmt, settings, 8)
    return ctx.time_hsvfilter_device(d, n, pitch, w, h, stride, fmt, settings, iters)
# warm the clocks
for _ in range(5): t(bufs[0], st)
def refill():
    # the H2D copies leave the device idle long enough to drop its clocks: put it back to work before timing
    for d in bufs: ctx.h2d(d, fr)
    for _ in range(6): ctx.time_hsvfilter_device(bufs[3], B, W*H*4, W, H, W*4, "xBGR", st, 40)
    ctx.h2d(bufs[3], fr)
print("== blocks/CU (8x4K RGBA hue90, ms per launch, frac of 8 TB/s)")
for tab in (1,):
    for bpc in (8, 16, 24, 32, 48, 64, 96, 128, 256):
        ctx.set_flag(mi355fx.FLAG_HSV_BLOCKS_PER_CU, bpc)
        refill()
        ms = min(t(bufs[i % 3], st) for i in range(3))
        print("blocks/CU %4d  %.4f ms  %.3f" % (bpc, ms, 2 * fr.nbytes / ms / 1e6 / 8000), flush=True)
ctx.set_flag(mi355fx.FLAG_HSV_BLOCKS_PER_CU, 64)
for tab in (1,):
    print("== settings / formats")
    for name, s in [("defaults", synth.HSV_SETTINGS["defaults"]), ("hue90", st), ("mixed", synth.HSV_SETTINGS["mixed"]), ("neg", (-200.25, 0.8, 0.1, 1.1, -0.05)),
                    ("wide+", (725.5, 1, 0, 1, 0)), ("wide-", (-1000.5, 1.1, 0, 0.9, 0)), ("generic inf", (float("inf"), 1, 0, 1, 0)), ("generic 1e7", (1e7, 1, 0, 1, 0))]:
        refill()
        ms = min(t(bufs[i % 4], s) for i in range(2))
        print("%-12s RGBA  %.4f ms  %.3f" % (name, ms, 2 * fr.nbytes / ms / 1e6 / 8000), flush=True)
    for fmt in ("BGRx", "xRGB", "ABGR"):
        refill()
        ms = min(t(bufs[i % 4], st, fmt) for i in range(2))
        print("%-12s %s  %.4f ms  %.3f" % ("hue90", fmt, ms, 2 * fr.nbytes / ms / 1e6 / 8000), flush=True)
# 3-byte formats: 8 x 4K RGB = 8 x 24.9 MB, the natural-like frame without its alpha byte (not the RGBA bytes reinterpreted:
# that would make every fourth "channel" random)
n3 = W * H * 3
rgb = np.ascontiguousarray(np.stack([one.reshape(H, W, 4)[..., :3]] * B)).reshape(-1)
for fmt in ("RGB", "BGR"):
    ctx.h2d(bufs[0], rgb)
    for _ in range(6): ctx.time_hsvfilter_device(bufs[3], B, W*H*4, W, H, W*4, "xBGR", st, 40)
    ms = min(t(bufs[0], st, fmt, pitch=n3, stride=W * 3) for i in range(3))
    print("hue90 %s 8x4K  %.4f ms  %.3f" % (fmt, ms, 2 * 8 * n3 / ms / 1e6 / 8000), flush=True)
# padded rows: 4K RGBA with stride 15,424 (64 B of padding per row), 8 frames
stride = W * 4 + 64
pitch = stride * H
dpad = ctx.alloc(pitch * B)
ctx.h2d(dpad, np.zeros(pitch * B, np.uint8))
ms = min(t(dpad, st, "RGBA", pitch=pitch, stride=stride) for i in range(2))
print("hue90 RGBA padded rows 8x4K  %.4f ms  %.3f" % (ms, 2 * fr.nbytes / ms / 1e6 / 8000), flush=True)
# config 2: 1080p BGRx, 32 frames per launch
w2, h2, n2 = 1920, 1080, 32
d2 = ctx.alloc(w2 * h2 * 4 * n2)
ctx.h2d(d2, np.stack([synth.smooth_frame(w2, h2)] * n2))
ms = min(t(d2, st, "BGRx", pitch=w2 * h2 * 4, w=w2, h=h2, stride=w2 * 4, n=n2) for i in range(3))
print("config 2: hue90 BGRx 32x1080p  %.4f ms  %.3f  %.0f frames/s" % (ms, 2 * w2 * h2 * 4 * n2 / ms / 1e6 / 8000, n2 / ms * 1e3), flush=True)
