"""How much of Dssim's time goes into tiles that touch the image border (their passes replicate edges per tap instead of reading constant
offsets)? create_image + compare_frames on frames of the SAME pixel count whose tiles are nearly all interior (4096 x 2176) or all border
(64 x 139264: two tile columns, both touch an edge). Run on the GPU box: python tools/dssim_border_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx

ctx = mi355fx.Context(0)
rng = np.random.default_rng(0)
for w, h in ((4096, 2176), (64, 139264), (3840, 2160), (1920, 1080), (960, 540), (480, 270)):
    a = rng.integers(0, 256, size=(h, w * 4), dtype=np.uint8); a[:, 3::4] = 255
    b = np.clip(a.astype(int) + rng.integers(-9, 10, a.shape), 0, 255).astype(np.uint8); b[:, 3::4] = 255
    da, db = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
    ctx.h2d(da, a.reshape(-1)); ctx.h2d(db, b.reshape(-1))
    def pair():
        x = ctx.dssim_create_image_device(da, w * 4, w, h)
        v = ctx.dssim_compare_frames_device(x, [db], w * 4, w, h)[0]
        ctx.dssim_free_image(x)
        return v
    for _ in range(5): pair()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): pair()
    dt = (time.perf_counter() - t0) / n
    tiles_x, tiles_y = (w + 31) // 32, (h + 15) // 16
    border = sum(1 for ty in range(tiles_y) for tx in (0, tiles_x - 1)) if tiles_x <= 2 else 0
    print("%6d x %-6d  %.4f ms per pair  %.2f ns per pixel  (tiles %d x %d)" % (w, h, dt * 1e3, dt * 1e9 / (w * h), tiles_x, tiles_y), flush=True)
    ctx.free(da); ctx.free(db)
ctx.close()
