"""BASELINE config 5 with the SSIM engine: videocompare hash-algo=dssim on 4K RGBA, frames resident in HBM:
time to build one DssimImage (the 'hash' of a frame) and to compare two; comparisons/s for a stream that hashes
reference + secondary every frame. CPU restatement (numpy) timed on a 1080p pair for scale."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx


def main():
    w, h = 3840, 2160
    rng = np.random.default_rng(0)
    a = np.kron(rng.integers(0, 256, (h // 8, w // 8, 4), dtype=np.uint8), np.ones((8, 8, 1), np.uint8)).reshape(h, w * 4); a[:, 3::4] = 255
    b = np.clip(a.astype(int) + rng.integers(-10, 11, a.shape), 0, 255).astype(np.uint8); b[:, 3::4] = 255
    ctx = mi355fx.Context(0)
    da, db = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
    ctx.h2d(da, a.reshape(-1)); ctx.h2d(db, b.reshape(-1))
    def one():
        x = ctx.dssim_create_image_device(da, w * 4, w, h); ctx.dssim_free_image(x)
    mi355fx.warm_clocks(one, ctx.synchronize)
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        x = ctx.dssim_create_image_device(da, w * 4, w, h); ctx.dssim_free_image(x)
    t_create = (time.perf_counter() - t0) / n
    x, y = ctx.dssim_create_image_device(da, w * 4, w, h), ctx.dssim_create_image_device(db, w * 4, w, h)
    t0 = time.perf_counter()
    for _ in range(n):
        d = ctx.dssim_compare(x, y)
    t_cmp = (time.perf_counter() - t0) / n
    for _ in range(10):
        ctx.dssim_compare_frames_device(x, [db], w * 4, w, h)
    t0 = time.perf_counter()
    for _ in range(n):
        d1 = ctx.dssim_compare_frames_device(x, [db], w * 4, w, h)[0]
    t_fused = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(n // 8):
        d8 = ctx.dssim_compare_frames_device(x, [db] * 8, w * 4, w, h)
    t_fused8 = (time.perf_counter() - t0) / (n // 8) / 8
    assert d1 == d and d8 == [d] * 8, (d, d1, d8)
    ctx.dssim_free_image(x); ctx.dssim_free_image(y)
    out = {"config": "videocompare hash-algo=dssim, 3840x2160 RGBA, device-resident frames", "dssim": d,
           "create_image_ms": t_create * 1e3, "compare_ms": t_cmp * 1e3,
           "comparisons_per_s_per_stream_pair": 1.0 / (2 * t_create + t_cmp),
           "hash_and_compare_ms": t_fused * 1e3, "hash_and_compare_ms_in_calls_of_8": t_fused8 * 1e3,
           "comparisons_per_s_pair_fused": 1.0 / (t_create + t_fused), "comparisons_per_s_pads_of_one_reference": 1.0 / (t_create / 8 + t_fused8)}
    from oracle import dssim_restate as D
    ws, hs = 1920, 1080
    sa, sb = a[:hs, : ws * 4].copy(), b[:hs, : ws * 4].copy()
    t0 = time.perf_counter()
    dd = D.compare(D.DssimImage(sa, ws, hs, ws * 4, 4), D.DssimImage(sb, ws, hs, ws * 4, 4))
    out["cpu_numpy_restatement_1080p_comparison_s"] = time.perf_counter() - t0
    print(json.dumps(out))
    ctx.free(da); ctx.free(db); ctx.close()


if __name__ == "__main__":
    main()
