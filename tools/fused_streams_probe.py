"""The fused hsvfilter -> colorlut launch (8 x 4K per launch, pristine batches from HBM) issued on ONE stream against the same launches
dealt out to K contexts (K streams, one composed table between them, no ordering between the streams until the end): does the next
launch fill the tail of a persistent 256-block kernel and cover the launch boundary? frames/s over STEPS x 4 launches.
Run on the GPU box: python tools/fused_streams_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

W, H, N = bench.W, bench.H, 8
STEPS = int(os.environ.get("STEPS", "20"))


def main():
    dev = torch.device("cuda:0")
    lut = parse_cube(synth.cube_text_3d(33))
    st = synth.HSV_SETTINGS["hue90"]
    pitch = W * H * 4
    pool = bench.SourcePool(torch, synth, dev, N, "smooth")
    master = [pool.new(k) for k in range(8)]
    n_launches = 4 * STEPS
    dst = [torch.empty_like(master[0]) for _ in range(4)]
    ctxs, streams = [], []
    for k in range(int(os.environ.get("K", "3"))):
        c = mi355fx.Context(0)
        s = torch.cuda.Stream(device=dev)
        c.set_stream(s.cuda_stream)
        c.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
        if os.environ.get("VARIANT"):
            c.set_flag(mi355fx.FLAG_LUT_VARIANT, int(os.environ["VARIANT"]))
        ctxs.append(c); streams.append(s)

    def warm(cs):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            for c in cs:
                for j in range(8):
                    c.hsv_colorlut_frames_device(master[j].data_ptr(), pitch, W * 4, dst[j % 4].data_ptr(), pitch, W * 4, N, W, H, st)
            torch.cuda.synchronize()

    for k_used in (1, 2, 3)[: len(ctxs)]:
        cs = ctxs[:k_used]
        for rep in range(3):
            warm(cs)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_launches):
                c = cs[i % k_used]
                c.hsv_colorlut_frames_device(master[i % 8].data_ptr(), pitch, W * 4, dst[i % 4].data_ptr(), pitch, W * 4, N, W, H, st)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("%d stream(s): %8.0f frames/s  (%s)" % (k_used, n_launches * N / dt, cs[0].colorlut_kernel_name()), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
