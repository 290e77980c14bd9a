"""A handful of 4K Dssim hashes / comparisons (profiling driver for tools/pmc_dssim.sh)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx
w, h = 3840, 2160
rng = np.random.default_rng(0)
a = np.kron(rng.integers(0, 256, (h // 8, w // 8, 4), dtype=np.uint8), np.ones((8, 8, 1), np.uint8)).reshape(h, w * 4); a[:, 3::4] = 255
b = np.clip(a.astype(int) + rng.integers(-10, 11, a.shape), 0, 255).astype(np.uint8); b[:, 3::4] = 255
ctx = mi355fx.Context(0)
da, db = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
ctx.h2d(da, a.reshape(-1)); ctx.h2d(db, b.reshape(-1))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
x = ctx.dssim_create_image_device(da, w * 4, w, h)
for _ in range(n):
    y = ctx.dssim_create_image_device(db, w * 4, w, h)
    v = ctx.dssim_compare(x, y)
    ctx.dssim_free_image(y)
    v2 = ctx.dssim_compare_frames_device(x, [db], w * 4, w, h)[0]
    assert v == v2
print("dssim", v)
