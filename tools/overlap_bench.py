"""Does hsvfilter of batch n+1 overlap colorlut of batch n when the two elements run on their own streams?
Compares one stream (serial) with two streams + events. 8 x 4K RGBA per batch, ring of 3 batches.
Measured with the interpolating colorlut kernel: 23.5 k frames/s either way - that kernel (1024 threads x 121 VGPRs) owns
every SIMD's register file, so no hsvfilter wave can be co-resident; a build limited to 96 VGPRs (amdgpu_waves_per_eu(5,5))
spills and drops to 14.8 k. With the table kernel (small blocks, co-residency possible): 36.1-42.8 k serial against 33.1-36.0 k
on two streams - hsvfilter of batch n+1 streams 0.5 GB through the Infinity Cache while colorlut of batch n is reading the
batch hsvfilter left there, and the lost cache hits cost more than the overlap gains."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube

W, H, N, R = 3840, 2160, 8, 3


def main():
    dev = torch.device("cuda", 0)
    lut = parse_cube(synth.cube_text_3d(33))
    st = synth.HSV_SETTINGS["hue90"]
    frame = torch.from_numpy(np.stack([synth.smooth_frame(W, H, seed=3 + i) for i in range(N)])).to(dev)
    bufs = [frame.clone() for _ in range(R)]
    outs = [torch.empty_like(frame) for _ in range(R)]
    pitch = W * H * 4
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    ca, cb = mi355fx.Context(0), mi355fx.Context(0)
    ca.set_stream(sa.cuda_stream); cb.set_stream(sb.cuda_stream)
    cb.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    ca.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    res = {}
    steps = 60

    def serial(variant):
        ca.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
        for k in range(steps + 5):
            if k == 5:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            b, o = bufs[k % R], outs[k % R]
            ca.hsvfilter_frames_device(b.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
            ca.colorlut_frames_device(b.data_ptr(), pitch, W * 4, o.data_ptr(), pitch, W * 4, N, W, H, "RGBA")
        torch.cuda.synchronize()
        return steps * N / (time.perf_counter() - t0)

    def overlapped(variant):
        cb.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
        ev_a = [torch.cuda.Event() for _ in range(steps + 5)]
        ev_b = [torch.cuda.Event() for _ in range(steps + 5)]
        for k in range(steps + 5):
            if k == 5:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            b, o = bufs[k % R], outs[k % R]
            if k >= R:
                sa.wait_event(ev_b[k - R])          # colorlut of the batch that used this buffer is done
            ca.hsvfilter_frames_device(b.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
            ev_a[k].record(sa)
            sb.wait_event(ev_a[k])
            cb.colorlut_frames_device(b.data_ptr(), pitch, W * 4, o.data_ptr(), pitch, W * 4, N, W, H, "RGBA")
            ev_b[k].record(sb)
        torch.cuda.synchronize()
        return steps * N / (time.perf_counter() - t0)

    for rep in range(2):
        res["serial_default_fps_%d" % rep] = serial(0)
        res["two_streams_default_fps_%d" % rep] = overlapped(0)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
