"""BASELINE config 5 shape on one GPU: 32 concurrent 4K RGBA streams, one comparison (two frame hashes) per stream
per frame time; frames resident in HBM. Reports comparisons/s, HBM GB/s vs the 8 TB/s peak and the CPU oracle rate."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx


def main():
    streams, w, h = 32, 3840, 2160
    n = 2 * streams
    ctx = mi355fx.Context(0)
    rng = np.random.default_rng(0)
    one = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
    d = ctx.alloc(n * one.nbytes)
    for k in range(n):
        ctx.h2d(d + k * one.nbytes, np.roll(one, k * 64, axis=1).reshape(-1))
    mi355fx.warm_clocks(lambda: ctx.videocompare_hash_frames_device(d, one.nbytes, w * 4, n, w, h), ctx.synchronize)
    iters = 100
    t0 = time.perf_counter()
    for _ in range(iters):
        hs = ctx.videocompare_hash_frames_device(d, one.nbytes, w * 4, n, w, h)
        dist = [ctx.videocompare_distance(hs[k], hs[streams + k]) for k in range(streams)]
    dt = (time.perf_counter() - t0) / iters
    out = {"config": "videocompare blockhash, %d streams x (reference + secondary) 4K RGBA frames per step" % streams,
           "comparisons_per_s": streams / dt, "ms_per_step": dt * 1e3, "bytes_per_step": n * one.nbytes,
           "GBps": n * one.nbytes / dt / 1e9, "frac_of_hbm_peak": n * one.nbytes / dt / 8e12}
    from oracle import oracle as O
    t0 = time.perf_counter()
    O.blockhash(one, w, h, w * 4, 4); O.blockhash(one, w, h, w * 4, 4)
    out["cpu_oracle_comparisons_per_s_1core"] = 1.0 / (time.perf_counter() - t0)
    print(json.dumps(out))
    ctx.free(d); ctx.close()


if __name__ == "__main__":
    main()
