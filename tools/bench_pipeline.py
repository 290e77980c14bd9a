"""Host-buffer (PCIe-inclusive) throughput of the hsvfilter!colorlut chain on 4K RGBA frames:
  (a) synchronous entry points on pageable buffers (two elements, two round trips per frame),
  (b) synchronous fused entry on pinned buffers, (c) mi355_pipe_* depth 1/2/3/4 on pinned buffers.
Never bench.py's `value` (that one is HBM-resident); reported in DESIGN.md."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H = 3840, 2160


def main():
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    st = synth.HSV_SETTINGS["hue90"]
    frame = synth.smooth_frame(W, H).reshape(-1)
    nbytes = frame.nbytes
    out = {}
    # (a) two synchronous element calls on pageable memory
    a, b = frame.copy(), np.zeros_like(frame)
    for _ in range(3):
        ctx.hsvfilter_frame_ip(a, W, W * 4, "RGBA", st); ctx.colorlut_frame(a, W * 4, b, W * 4, W, H)
    n = 40
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.hsvfilter_frame_ip(a, W, W * 4, "RGBA", st); ctx.colorlut_frame(a, W * 4, b, W * 4, W, H)
    out["sync_two_elements_pageable_fps"] = n / (time.perf_counter() - t0)
    # (c) pipeline, pinned
    ring = 8
    srcs = [ctx.host_array(nbytes) for _ in range(ring)]
    dsts = [ctx.host_array(nbytes) for _ in range(ring)]
    for s in srcs:
        s[:] = frame
    for depth in (1, 2, 3, 4):
        pipe = ctx.pipe_create(depth, nbytes)
        for k in range(ring):
            ctx.pipe_submit_hsv_colorlut(pipe, srcs[k], W * 4, dsts[k], W * 4, W, H, st)
        ctx.pipe_wait_all(pipe)
        n = 200
        t0 = time.perf_counter()
        for k in range(n):
            ctx.pipe_submit_hsv_colorlut(pipe, srcs[k % ring], W * 4, dsts[k % ring], W * 4, W, H, st)
        ctx.pipe_wait_all(pipe)
        dt = time.perf_counter() - t0
        out["pipe_depth%d_pinned_fps" % depth] = n / dt
        out["pipe_depth%d_pcie_GBps_each_way" % depth] = n * nbytes / dt / 1e9
        ctx.pipe_destroy(pipe)
    # pageable buffers through the pipeline
    pipe = ctx.pipe_create(3, nbytes)
    pa = [frame.copy() for _ in range(4)]; pb = [np.zeros_like(frame) for _ in range(4)]
    n = 40
    t0 = time.perf_counter()
    for k in range(n):
        ctx.pipe_submit_hsv_colorlut(pipe, pa[k % 4], W * 4, pb[k % 4], W * 4, W, H, st)
    ctx.pipe_wait_all(pipe)
    out["pipe_depth3_pageable_fps"] = n / (time.perf_counter() - t0)
    ctx.pipe_destroy(pipe)
    print(json.dumps(out))
    for a_ in srcs + dsts:
        ctx.host_free(a_)
    ctx.close()


if __name__ == "__main__":
    main()
