"""SURVEY.md 8(d) timing protocol: single-stream vs N-concurrent-stream (one mi355_ctx + hipStream per GStreamer element
instance) throughput of the hsvfilter -> colorlut chain, one 4K RGBA frame per launch (what an unmodified per-buffer
pipeline issues), frames resident in HBM. Compare with bench.py, which batches 8 frames of ONE stream per launch."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H = 3840, 2160


def run(n_streams, frames_per_stream, fused):
    lut = parse_cube(synth.cube_text_3d(33))
    st = synth.HSV_SETTINGS["hue90"]
    frame = synth.smooth_frame(W, H).reshape(-1)
    ctxs, bufs = [], []
    for _ in range(n_streams):
        c = mi355fx.Context(0)
        c.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        a, b = c.alloc(frame.nbytes), c.alloc(frame.nbytes)
        c.h2d(a, frame)
        ctxs.append(c); bufs.append((a, b))

    def step():
        for c, (a, b) in zip(ctxs, bufs):
            if fused:
                c.hsv_colorlut_frames_device(a, frame.nbytes, W * 4, b, frame.nbytes, W * 4, 1, W, H, st)
            else:
                c.hsvfilter_frames_device(a, 1, frame.nbytes, W, H, W * 4, "RGBA", st)
                c.colorlut_frames_device(a, frame.nbytes, W * 4, b, frame.nbytes, W * 4, 1, W, H, "RGBA")
    def sync_all():
        for c in ctxs:
            c.synchronize()
    mi355fx.warm_clocks(step, sync_all)
    for _ in range(3):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(frames_per_stream):
        step()
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    for c, (a, b) in zip(ctxs, bufs):
        c.free(a); c.free(b); c.close()
    return n_streams * frames_per_stream / dt


def main():
    out = {}
    for n in (1, 8, 32):
        out["two_kernels_%d_streams_fps" % n] = run(n, 256, False)
        out["fused_%d_streams_fps" % n] = run(n, 256, True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
