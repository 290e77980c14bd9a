"""sofalizer (uniformly partitioned FFT convolution, mi355_sofa_*): blocks per second and real-time factor for the element's
default geometry (partition-length 64, block-length 256, audio/hrtf/src/sofa/imp.rs:37-41) and larger ones, device-resident
input and host buffers. Run on the GPU box: python tools/bench_sofa.py"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd")); sys.path.insert(0, ROOT)
import mi355fx

RATE = 48000


def run(channels, taps, part, block, blocks=300):
    ctx = mi355fx.Context(0)
    rng = np.random.default_rng(1)
    ctx.sofa_setup(channels, taps, part, block)
    for c in range(channels):
        h = (rng.standard_normal((2, taps)) * np.exp(-np.arange(taps) / (0.2 * taps))).astype(np.float32)
        ctx.sofa_set_filter(c, h[0], h[1], 0, 0)
    x = rng.uniform(-1, 1, (block, channels)).astype(np.float32)
    gains = np.full(channels, 0.5, np.float32)
    d_in, d_out = ctx.alloc(x.nbytes), ctx.alloc(block * 8)
    ctx.h2d(d_in, x.reshape(-1))
    for _ in range(50): ctx.sofa_process_block_device(d_in, d_out, gains)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(blocks): ctx.sofa_process_block_device(d_in, d_out, gains)
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / blocks
    t0 = time.perf_counter()
    for _ in range(blocks // 4): ctx.sofa_process_block(x, gains)
    dth = (time.perf_counter() - t0) / (blocks // 4)
    ctx.close()
    return {"config": "sofalizer %d channels, %d-tap filters, partition %d, block %d, %d Hz" % (channels, taps, part, block, RATE),
            "device_ms_per_block": dt * 1e3, "realtime_factor": (block / RATE) / dt, "host_buffer_ms_per_block": dth * 1e3,
            "host_buffer_realtime_factor": (block / RATE) / dth}


if __name__ == "__main__":
    for channels, taps, part, block in ((6, 512, 64, 256), (64, 512, 64, 256), (64, 2048, 256, 1024), (64, 4096, 512, 4096)):
        print(json.dumps(run(channels, taps, part, block)), flush=True)
