"""Does a hipGraph close the launch boundaries of the headline chain? One step of the bench = 4 batches x (hsvfilter in place, colorlut) =
8 launches on one stream. Direct: the eight library calls per step, as bench.py issues them. Graph: the same eight calls captured ONCE
(table kernel pinned: the measured choice records events, which a capture cannot hold) and replayed with hipGraphLaunch. Same buffers
every step in both forms (a graph bakes its pointers in; the bench's pristine batches would need hipGraphExecKernelNodeSetParams per
node and step). Wall clock over 20 steps, best of 5. Run on the GPU box: python tools/graph_probe.py"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N, PAIRS, STEPS = 3840, 2160, 8, 4, 20
FB = W * H * 4


def main():
    hip = C.CDLL("libamdhip64.so")
    ctx = mi355fx.Context(0)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, 5)     # the memoised table through the gather kernel, pinned
    st = synth.HSV_SETTINGS["hue90"]
    frames = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(-1)
    src = [ctx.alloc(N * FB) for _ in range(PAIRS)]
    dst = [ctx.alloc(N * FB) for _ in range(PAIRS)]
    for s in src:
        ctx.h2d(s, frames)
    stream = C.c_void_p(ctx.stream)

    def step():
        for k in range(PAIRS):
            ctx.hsvfilter_frames_device(src[k], N, FB, W, H, W * 4, "RGBA", st)
            ctx.colorlut_frames_device(src[k], FB, W * 4, dst[k], FB, W * 4, N, W, H, "RGBA")

    for _ in range(10):
        step()
    ctx.synchronize()

    def timed(fn):
        best = 1e9
        for _ in range(5):
            for _ in range(3):
                fn()
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(STEPS):
                fn()
            ctx.synchronize()
            best = min(best, (time.perf_counter() - t0) / STEPS * 1e3)
        return best

    t_direct = timed(step)
    graph, exe = C.c_void_p(), C.c_void_p()
    rc = hip.hipStreamBeginCapture(stream, 0)   # hipStreamCaptureModeGlobal
    assert rc == 0, rc
    step()
    rc = hip.hipStreamEndCapture(stream, C.byref(graph))
    assert rc == 0 and graph.value, rc
    rc = hip.hipGraphInstantiate(C.byref(exe), graph, None, None, 0)
    assert rc == 0, rc
    t_graph = timed(lambda: hip.hipGraphLaunch(exe, stream))
    # one graph for all twenty steps (160 kernel nodes)
    graph20, exe20 = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0
    for _ in range(STEPS):
        step()
    assert hip.hipStreamEndCapture(stream, C.byref(graph20)) == 0
    assert hip.hipGraphInstantiate(C.byref(exe20), graph20, None, None, 0) == 0
    best = 1e9
    for _ in range(5):
        hip.hipGraphLaunch(exe20, stream); ctx.synchronize()
        t0 = time.perf_counter()
        hip.hipGraphLaunch(exe20, stream); ctx.synchronize()
        best = min(best, (time.perf_counter() - t0) / STEPS * 1e3)
    fps = lambda ms: N * PAIRS / ms * 1e3
    print("8 x 4K x 4 batches per step, hsvfilter + colorlut (table gather kernel pinned), same buffers every step")
    print("  direct library calls          %.4f ms per step  %.0f frames/s" % (t_direct, fps(t_direct)))
    print("  one graph per step, replayed  %.4f ms per step  %.0f frames/s  (%+.1f %%)" % (t_graph, fps(t_graph), 100 * (t_direct / t_graph - 1)))
    print("  one graph of twenty steps     %.4f ms per step  %.0f frames/s  (%+.1f %%)" % (best, fps(best), 100 * (t_direct / best - 1)))
    for p in src + dst:
        ctx.free(p)
    ctx.close()


if __name__ == "__main__":
    main()
