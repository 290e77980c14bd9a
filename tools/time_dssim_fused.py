"""Time mi355_dssim_compare_frames_device on 4K frames (calls of 8); results are not checked (used with experimental builds)."""
import os, sys, time
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
x = ctx.dssim_create_image_device(da, w * 4, w, h)
for _ in range(20):
    ctx.dssim_compare_frames_device(x, [db] * 8, w * 4, w, h)
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        v = ctx.dssim_compare_frames_device(x, [db] * 8, w * 4, w, h)
    best = min(best, (time.perf_counter() - t0) / 80)
print(os.environ.get("MI355FX_LIB", "default"), "hash+compare %.4f ms per frame, dssim %r" % (best * 1e3, v[0]))
