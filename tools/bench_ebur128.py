"""ebur128level meter on the device: real-time factor for one 48 kHz stream (host buffers, 100 ms buffers as a live
pipeline delivers them) in the element's default mode (all measurements incl. true peak) and without true peak."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx


def main():
    rate, seconds = 48000, 60.0
    out = {}
    ctx = mi355fx.Context(0)
    for ch in (2, 6):
        t = np.arange(int(seconds * rate)) / rate
        x = np.stack([0.1 * np.sin(2 * np.pi * (220 + 5 * c) * t) for c in range(ch)], 1).astype(np.float32).reshape(-1)
        for name, mode in (("all_modes", 63), ("without_true_peak", 31)):
            ctx.ebur128_setup(ch, rate, mode)
            step = rate // 10 * ch
            t0 = time.perf_counter()
            for k in range(0, x.size, step):
                ctx.ebur128_add_frames(x[k:k + step])
            g = ctx.ebur128_loudness_global()
            dt = time.perf_counter() - t0
            out["%dch_%s_realtime_factor" % (ch, name)] = seconds / dt
            out["%dch_%s_global_LUFS" % (ch, name)] = g
    # many streams at once (mi355_ebur128_*_batch): S stereo 48 kHz streams, one second of audio per call, all modes.
    # The serial K-weighting recurrence of one stream cannot be split; S streams are S independent recurrences.
    for S in (32, 256, 1024):
        secs = 10
        t = np.arange(rate) / rate
        one = np.stack([0.1 * np.sin(2 * np.pi * 330 * t), 0.1 * np.sin(2 * np.pi * 331 * t)], 1).astype(np.float32)
        batch = np.broadcast_to(one, (S,) + one.shape).copy()
        ctx.ebur128_setup_batch(S, 2, rate, 63)
        ctx.ebur128_add_frames_batch(batch)
        ctx.ebur128_reset()
        t0 = time.perf_counter()
        for _ in range(secs):
            ctx.ebur128_add_frames_batch(batch)
        g = ctx.ebur128_loudness_batch(2)
        dt = time.perf_counter() - t0
        out["batch_%d_stereo_streams_realtime_factor_per_stream" % S] = secs / dt
        out["batch_%d_stereo_streams_aggregate_stream_seconds_per_s" % S] = S * secs / dt
        assert np.all(g == g[0])
    from oracle import oracle as O
    ms = O.EbuR128(2, rate)
    xs = np.tile(one.reshape(-1), 30)
    t0 = time.perf_counter(); ms.add_frames(xs); dt = time.perf_counter() - t0
    out["cpu_oracle_stereo_all_modes_realtime_factor_1_core"] = 30.0 / dt
    out["cpu_oracle_stereo_aggregate_on_%d_cores_if_perfectly_parallel" % (os.cpu_count() or 1)] = 30.0 / dt * (os.cpu_count() or 1)
    m = O.EbuR128(6, rate)
    t0 = time.perf_counter(); m.add_frames(x); out["cpu_oracle_6ch_all_modes_realtime_factor"] = seconds / (time.perf_counter() - t0)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
