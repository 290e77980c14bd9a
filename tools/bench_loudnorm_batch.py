"""audioloudnorm as a batch of streams in lock step (mi355_loudnorm_*_batch, round 3): n stereo 192 kHz streams, inner frames of
100 ms from device-resident buffers, against one context per stream and the C oracle on one core. Prints the aggregate
real-time factor (stream-seconds of audio per second). Run on the GPU box: python tools/bench_loudnorm_batch.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx

RATE, CH, FRAME = 192000, 2, 19200


def material(seconds, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * RATE)) / RATE
    x = np.stack([0.05 * np.sin(2 * np.pi * (440.0 + 3 * c + seed) * t) for c in range(CH)], 1)
    for _ in range(int(seconds * 2)):
        i = int(rng.uniform(0, seconds - 0.02) * RATE)
        x[i:i + 400] *= rng.uniform(10, 40)
    return x


def main():
    seconds = float(os.environ.get("SECONDS_AUDIO", "8"))
    for S in [int(v) for v in os.environ.get("STREAMS", "32,256").split(",")]:
        ctx = mi355fx.Context(0)
        base = [material(seconds, s % 8) for s in range(min(S, 8))]
        n = len(base[0])
        ctx.loudnorm_setup_batch(S, CH)
        first = 3 * RATE
        x0 = np.stack([base[s % len(base)][:first].reshape(-1) for s in range(S)])
        d_in, d_out = ctx.alloc(x0.nbytes), ctx.alloc(S * FRAME * CH * 8 * 31)
        ctx.h2d(d_in, x0)
        ctx.loudnorm_process_batch_device(d_in, first * CH, first, d_out, FRAME * CH, FRAME)
        ctx.synchronize()
        inner = (n - first) // FRAME
        frames = [np.stack([base[s % len(base)][first + k * FRAME: first + (k + 1) * FRAME].reshape(-1) for s in range(S)]) for k in range(inner)]
        d_frames = []
        for f in frames:
            d = ctx.alloc(f.nbytes); ctx.h2d(d, f); d_frames.append(d)
        ctx.synchronize()
        t0 = time.perf_counter()
        for d in d_frames:
            ctx.loudnorm_process_batch_device(d, FRAME * CH, FRAME, d_out, FRAME * CH, FRAME)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        audio = inner * FRAME / RATE
        print("batch of %4d streams, device-resident frames: %.3f s for %.1f s of audio per stream -> %.0fx real time aggregate (%.1fx per stream)"
              % (S, dt, audio, S * audio / dt, audio / dt), flush=True)
        # the same through host buffers (PCIe both ways)
        ctx.loudnorm_teardown(); ctx.loudnorm_setup_batch(S, CH)
        ctx.loudnorm_process_batch(x0)
        t0 = time.perf_counter()
        for f in frames:
            ctx.loudnorm_process_batch(f)
        dt = time.perf_counter() - t0
        print("batch of %4d streams, host buffers:            %.3f s -> %.0fx real time aggregate" % (S, dt, S * audio / dt), flush=True)
        ctx.loudnorm_teardown()
        ctx.close()
    # one context per stream (round 2's form), and the oracle on one core
    ctx = mi355fx.Context(0)
    x = material(seconds, 0)
    ctx.loudnorm_setup(CH)
    t0 = time.perf_counter(); ctx.loudnorm_push(x); ctx.loudnorm_drain(); dt = time.perf_counter() - t0
    print("one single-stream context: %.3f s for %.1f s -> %.0fx real time" % (dt, seconds, seconds / dt))
    from oracle import oracle as O
    ln = O.LoudNorm(CH)
    t0 = time.perf_counter(); ln.push(x); ln.drain(); dt = time.perf_counter() - t0
    print("C oracle, one core: %.3f s -> %.0fx real time" % (dt, seconds / dt))


if __name__ == "__main__":
    main()
