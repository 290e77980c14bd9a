"""audioloudnorm on the device: real-time factor for one stereo 192 kHz f64 stream (host buffers in/out) and the
deviation from the CPU oracle. Run on the GPU box: python tools/bench_loudnorm.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx

RATE = 192000


def main():
    ch, seconds = 2, 30.0
    t = np.arange(int(seconds * RATE)) / RATE
    rng = np.random.default_rng(0)
    x = np.stack([0.05 * np.sin(2 * np.pi * 440 * t) * (1 + 0.5 * np.sin(2 * np.pi * 0.2 * t)), 0.05 * np.sin(2 * np.pi * 554 * t)], 1)
    for s in rng.uniform(3.2, seconds - 0.5, 40):
        i = int(s * RATE); x[i:i + int(rng.integers(10, 3000))] *= rng.uniform(10, 25)
    ctx = mi355fx.Context(0)
    ctx.loudnorm_setup(ch)
    outs = []
    t0 = time.perf_counter()
    for k in range(0, len(x), 19200 * 5):
        outs.append(ctx.loudnorm_push(x[k:k + 19200 * 5]))
    outs.append(ctx.loudnorm_drain())
    dt = time.perf_counter() - t0
    y = np.concatenate(outs)
    out = {"config": "audioloudnorm, 1 stream, %d ch f64 @192 kHz, %.0f s, 40 over-ceiling bursts" % (ch, seconds),
           "device_seconds": dt, "realtime_factor": seconds / dt, "peak_out": float(np.abs(y).max())}
    from oracle import oracle as O
    ln = O.LoudNorm(ch)
    t0 = time.perf_counter()
    e = [ln.push(x[k:k + 19200 * 5]) for k in range(0, len(x), 19200 * 5)] + [ln.drain()]
    out["cpu_oracle_seconds"] = time.perf_counter() - t0
    e = np.concatenate(e)
    out["max_rel_dev_vs_oracle"] = float(np.abs(y - e).max() / np.abs(e).max())
    out["bit_identical_fraction"] = float((y == e).mean())
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
