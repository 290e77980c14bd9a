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
    from oracle import oracle as O
    m = O.EbuR128(6, rate)
    t0 = time.perf_counter(); m.add_frames(x); out["cpu_oracle_6ch_all_modes_realtime_factor"] = seconds / (time.perf_counter() - t0)
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
