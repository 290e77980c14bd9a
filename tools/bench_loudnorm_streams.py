"""audioloudnorm as a deployment: S independent stereo 192 kHz f64 streams, one host thread + one mi355_ctx (own HIP stream,
own limiter state machine) per stream, all running concurrently (ctypes releases the GIL during the calls). One stream is a
serial recurrence the GPU runs slower than a CPU core; what the device offers is that the streams' kernels overlap.
Prints one JSON line: aggregate real-time factor per stream count. Run on the GPU box: python tools/bench_loudnorm_streams.py"""
import json, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx

RATE = 192000


def signal(seed, seconds, ch=2):
    t = np.arange(int(seconds * RATE)) / RATE
    rng = np.random.default_rng(seed)
    x = np.stack([0.05 * np.sin(2 * np.pi * (440 + seed) * t) * (1 + 0.5 * np.sin(2 * np.pi * 0.2 * t)), 0.05 * np.sin(2 * np.pi * 554 * t)], 1)
    for s in rng.uniform(3.2, seconds - 0.5, 10):
        i = int(s * RATE); x[i:i + int(rng.integers(10, 3000))] *= rng.uniform(10, 25)
    return x


def worker(k, x, barrier, out):
    ctx = mi355fx.Context(0)
    ctx.loudnorm_setup(2)
    barrier.wait()
    t0 = time.perf_counter()
    for j in range(0, len(x), 19200 * 5):
        ctx.loudnorm_push(x[j:j + 19200 * 5])
    ctx.loudnorm_drain()
    out[k] = time.perf_counter() - t0
    ctx.close()


def main():
    seconds = 10.0
    res = {}
    for S in [int(a) for a in sys.argv[1:]] or [1, 8, 32]:
        xs = [signal(k, seconds) for k in range(S)]
        out = [0.0] * S
        barrier = threading.Barrier(S + 1)
        th = [threading.Thread(target=worker, args=(k, xs[k], barrier, out)) for k in range(S)]
        for t in th: t.start()
        barrier.wait()
        t0 = time.perf_counter()
        for t in th: t.join()
        wall = time.perf_counter() - t0
        res["%d_streams_aggregate_realtime_factor" % S] = S * seconds / wall
        res["%d_streams_wall_s" % S] = wall
    from oracle import oracle as O
    x = signal(0, seconds)
    ln = O.LoudNorm(2)
    t0 = time.perf_counter()
    for j in range(0, len(x), 19200 * 5): ln.push(x[j:j + 19200 * 5])
    ln.drain()
    res["cpu_oracle_1core_realtime_factor"] = seconds / (time.perf_counter() - t0)
    res["config"] = "audioloudnorm, N concurrent streams (thread + context each), 2 ch f64 @192 kHz, %.0f s each" % seconds
    print(json.dumps(res))


if __name__ == "__main__":
    main()
