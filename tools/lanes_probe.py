"""The headline chain (hsvfilter in place -> colorlut, 8 x 4K per batch, every pair on a pristine batch) issued three ways:
the two element calls per pair from Python (what bench.py did up to round 4), mi355_hsv_colorlut_chain_batches_device with one
lane, and with two lanes (odd batches on a side stream). Frames/s over STEPS steps of 4 pairs, outputs compared.
Run on the GPU box: python tools/lanes_probe.py"""
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
    ctx = mi355fx.Context(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lut = parse_cube(synth.cube_text_3d(33))
    ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    st = synth.HSV_SETTINGS["hue90"]
    pitch = W * H * 4
    pool = bench.SourcePool(torch, synth, dev, N, "smooth")
    master = [pool.new(k) for k in range(8)]
    n_pairs = 4 * STEPS
    work = [torch.empty_like(master[0]) for _ in range(n_pairs)]
    dst = [torch.empty_like(master[0]) for _ in range(4)]

    def refill():
        for k, w_ in enumerate(work):
            w_.copy_(master[k % len(master)])

    def ramp():
        t0 = time.perf_counter()
        scratch = torch.empty_like(master[0])
        while time.perf_counter() - t0 < 0.3:
            for _ in range(8):
                scratch.copy_(master[0])
                ctx.hsvfilter_frames_device(scratch.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
                ctx.colorlut_frames_device(scratch.data_ptr(), pitch, W * 4, dst[0].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
            torch.cuda.synchronize()

    def run(mode):
        refill()
        ramp()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(STEPS):
            ws = work[4 * s:4 * s + 4]
            if mode == "python":
                for j, w_ in enumerate(ws):
                    ctx.hsvfilter_frames_device(w_.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
                    ctx.colorlut_frames_device(w_.data_ptr(), pitch, W * 4, dst[j].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
            else:
                ctx.chain_batches_device([w_.data_ptr() for w_ in ws], [d.data_ptr() for d in dst], N, pitch, W * 4, W, H, "RGBA", st, lanes=1 if mode == "lanes1" else 2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return n_pairs * N / dt, [d.clone() for d in dst], ctx.colorlut_kernel_name()

    for rep in range(int(os.environ.get("REPS", "3"))):
        ref = None
        for mode in ("python", "lanes1", "lanes2"):
            fps, out, name = run(mode)
            same = "" if ref is None else ("  outputs identical: %s" % all(bool((a == b).all()) for a, b in zip(ref, out)))
            if ref is None:
                ref = out
            print("%-7s %8.0f frames/s  (%s)%s" % (mode, fps, name, same), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
