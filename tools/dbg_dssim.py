import sys, os
sys.path.insert(0, "gst-plugins-rs_amd"); sys.path.insert(0, ".")
import numpy as np, mi355fx
ctx = mi355fx.Context(0)
for (w, h, ch, fmt) in ((35, 20, 3, "RGB"), (35, 20, 4, "RGBA"), (64, 48, 3, "RGB"), (322, 246, 4, "RGBA")):
    rng = np.random.default_rng(w * 7 + h)
    smooth = np.kron(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, ch), dtype=np.uint8), np.ones((8, 8, 1), np.uint8))[:h, :w].reshape(h, w * ch).copy()
    if ch == 4: smooth[:, 3::4] = 255
    for stride in (w * ch, w * ch + 12):
        p = np.zeros((h, stride), np.uint8); p[:, : w * ch] = smooth
        q = p.copy()
        a = ctx.dssim_create_image(p, stride, w, h, fmt)
        b = ctx.dssim_create_image(q, stride, w, h, fmt)
        v_alive = ctx.dssim_compare(a, b)
        c = ctx.dssim_create_image(p.copy(), stride, w, h, fmt)
        junk = np.full((h, stride), 77, np.uint8)
        v_temp = ctx.dssim_compare(a, c)
        d = ctx.alloc(p.nbytes); ctx.h2d(d, p.reshape(-1))
        e = ctx.dssim_create_image_device(d, stride, w, h, fmt)
        v_dev = ctx.dssim_compare(a, e)
        print(w, h, fmt, "stride", stride, "alive", v_alive, "temporary", v_temp, "device", v_dev)
        for im in (a, b, c, e): ctx.dssim_free_image(im)
        ctx.free(d)
