"""BASELINE config 5 deployment shape for the SSIM engine: N independent streams on one GPU, one host thread + one
mi355_ctx (own HIP stream, own image pools) per stream, each hashing reference + secondary 4K frames and comparing them,
all threads running concurrently (ctypes releases the GIL during the calls). Prints aggregate comparisons/s.
Run on the GPU box: python tools/bench_dssim_streams.py [streams ...]"""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx

W, H = 3840, 2160


def worker(k, a, b, iters, barrier, out):
    ctx = mi355fx.Context(0)
    da, db = ctx.alloc(a.nbytes), ctx.alloc(b.nbytes)
    ctx.h2d(da, a.reshape(-1)); ctx.h2d(db, b.reshape(-1))

    def one():
        x = ctx.dssim_create_image_device(da, W * 4, W, H)
        y = ctx.dssim_create_image_device(db, W * 4, W, H)
        d = ctx.dssim_compare(x, y)
        ctx.dssim_free_image(x); ctx.dssim_free_image(y)
        return d
    for _ in range(20):
        one()
    barrier.wait()
    t0 = time.perf_counter()
    for _ in range(iters):
        d = one()
    out[k] = (time.perf_counter() - t0, d)
    barrier.wait()
    ctx.free(da); ctx.free(db); ctx.close()


def main():
    rng = np.random.default_rng(0)
    a = np.kron(rng.integers(0, 256, (H // 8, W // 8, 4), dtype=np.uint8), np.ones((8, 8, 1), np.uint8)).reshape(H, W * 4); a[:, 3::4] = 255
    b = np.clip(a.astype(int) + rng.integers(-10, 11, a.shape), 0, 255).astype(np.uint8); b[:, 3::4] = 255
    res = {}
    for n in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8, 16, 32]:
        iters = max(20, 400 // n)
        barrier = threading.Barrier(n)
        out = [None] * n
        ts = [threading.Thread(target=worker, args=(k, a, b, iters, barrier, out)) for k in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        dt = max(o[0] for o in out)
        res["%d_streams_comparisons_per_s" % n] = n * iters / dt
        print(n, "streams:", round(n * iters / dt, 1), "comparisons/s  dssim", out[0][1], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
