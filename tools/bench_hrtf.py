"""BASELINE config 4: hrtfrender, 64 sources, 48 kHz f32 (44.1 kHz sphere mesh rate is rewritten to 48 kHz), block
512 x 8 steps. Reports device blocks/s with the input resident in HBM, the real-time factor, the host-buffer
(PCIe-inclusive) rate, and the CPU oracle (FFT overlap-save, 1 thread) on the same workload.
Run on the GPU box: python tools/bench_hrtf.py [--taps 256] [--sources 64]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import mi355fx
from mi355fx import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taps", type=int, default=256)
    ap.add_argument("--sources", type=int, default=64)
    ap.add_argument("--blocks", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--method", type=int, default=0, help="MI355_FLAG_HRTF_METHOD: 0 by HRIR length, 1 overlap-save FFT, 2 time-domain FIR")
    a = ap.parse_args()
    rate, steps, bl = 48000, 8, 512
    frames = steps * bl
    mesh = open(os.path.join(ROOT, "tests", "golden", "test.hrir"), "rb").read()
    data = synth.hrir_sphere_bytes(mesh, a.taps, rate=rate)
    ctx = mi355fx.Context(0)
    ctx.hrtf_load_sphere(data, rate)
    ctx.set_flag(mi355fx.FLAG_HRTF_METHOD, a.method)
    ctx.hrtf_setup(a.sources, bl, steps)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (frames, a.sources)).astype(np.float32)
    pos = rng.standard_normal((a.sources, 3)).astype(np.float32)
    gains = np.full(a.sources, 0.5, np.float32)
    d_in, d_out = ctx.alloc(x.nbytes), ctx.alloc(frames * 8)
    ctx.h2d(d_in, x.reshape(-1))
    mi355fx.warm_clocks(lambda: ctx.hrtf_process_block_device(d_in, d_out, pos, gains), ctx.synchronize)
    t0 = time.perf_counter()
    for i in range(a.blocks):
        pos[i % a.sources, 0] += 0.01
        ctx.hrtf_process_block_device(d_in, d_out, pos, gains)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(a.blocks // 4):
        ctx.hrtf_process_block(x, pos, gains)
    dth = (time.perf_counter() - t0) / (a.blocks // 4)
    out = {"config": "hrtfrender %d sources, %d-tap HRIRs, %d Hz f32, block %dx%d, method %d" % (a.sources, a.taps, rate, bl, steps, a.method),
           "device_blocks_per_s": a.blocks / dt, "device_ms_per_block": dt / a.blocks * 1e3,
           "realtime_factor": (frames / rate) / (dt / a.blocks),
           "host_buffer_ms_per_block": dth * 1e3, "host_buffer_realtime_factor": (frames / rate) / dth,
           "macs_per_block": frames * a.sources * 2 * a.taps}
    out["device_GMAC_per_s"] = out["macs_per_block"] / (dt / a.blocks) / 1e9
    if not a.no_cpu:
        from oracle import oracle as O
        sphere = O.HrirSphere(data, rate)
        r = O.HrtfRender(sphere, a.sources, steps, bl)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 8.0:
            r.process_block(x, pos, gains)
            n += 1
        dtc = (time.perf_counter() - t0) / n
        out["cpu_oracle_ms_per_block"] = dtc * 1e3
        out["cpu_oracle_realtime_factor"] = (frames / rate) / dtc
        out["cpu_note"] = "oracle C (generic mixed-radix f32 FFT, O(p^2) butterflies), 1 thread; not the reference's rustfft"
    print(json.dumps(out))
    ctx.free(d_in); ctx.free(d_out); ctx.close()


if __name__ == "__main__":
    main()
