"""hsvfilter and colorlut as TWO ELEMENTS with their own contexts and HIP streams - what `hsvfilter ! queue ! colorlut` is in GStreamer:
hsvfilter's stream filters batch k + 1 while colorlut's stream looks up batch k (ordered by one event per batch; hsvfilter never runs more
than AHEAD batches in front of colorlut, so that the intermediates stay in the Infinity Cache). hsvfilter is VALU-bound and colorlut's gather
kernel is bound by the texture-address path: the two use different parts of a CU. Against the same launches on ONE stream (bench.py's
headline), for N frames per launch. Frames/s over STEPS x 4 pairs, every pair on a pristine batch, outputs compared.
Run on the GPU box: python tools/pipeline2_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

W, H = bench.W, bench.H
STEPS = int(os.environ.get("STEPS", "20"))


def main():
    dev = torch.device("cuda:0")
    c_h, c_l = mi355fx.Context(0), mi355fx.Context(0)
    s_h, s_l = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    c_h.set_stream(s_h.cuda_stream)
    c_l.set_stream(s_l.cuda_stream)
    lut = parse_cube(synth.cube_text_3d(33))
    c_l.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
    st = synth.HSV_SETTINGS["hue90"]
    pitch = W * H * 4
    for N in [int(x) for x in os.environ.get("BATCHES", "8,4,2").split(",")]:
        pool = bench.SourcePool(torch, synth, dev, N, "smooth")
        master = [pool.new(k) for k in range(8)]
        n_pairs = 4 * STEPS * (8 // N)
        work = [torch.empty_like(master[0]) for _ in range(n_pairs)]
        dst = [torch.empty_like(master[0]) for _ in range(4)]
        evs = [torch.cuda.Event() for _ in range(n_pairs)]      # hsvfilter of pair k done
        evl = [torch.cuda.Event() for _ in range(n_pairs)]      # colorlut of pair k done

        def refill():
            for k, w_ in enumerate(work):
                w_.copy_(master[k % len(master)])

        def one_stream(n):
            with torch.cuda.stream(s_l):
                for k in range(n):
                    w_ = work[k % n_pairs]
                    c_l.hsvfilter_frames_device(w_.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
                    c_l.colorlut_frames_device(w_.data_ptr(), pitch, W * 4, dst[k % 4].data_ptr(), pitch, W * 4, N, W, H, "RGBA")

        def two_streams(n, ahead):
            for k in range(n):
                w_ = work[k % n_pairs]
                with torch.cuda.stream(s_h):
                    if k - ahead - 1 >= 0:
                        s_h.wait_event(evl[(k - ahead - 1) % n_pairs])     # not more than `ahead` batches in front of colorlut
                    c_h.hsvfilter_frames_device(w_.data_ptr(), N, pitch, W, H, W * 4, "RGBA", st)
                    evs[k % n_pairs].record(s_h)
                with torch.cuda.stream(s_l):
                    s_l.wait_event(evs[k % n_pairs])
                    c_l.colorlut_frames_device(w_.data_ptr(), pitch, W * 4, dst[k % 4].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
                    evl[k % n_pairs].record(s_l)

        def measure(fn):
            refill()
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 0.3:       # ramp (the sources degrade; refilled below)
                fn(16)
                torch.cuda.synchronize()
            refill()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn(n_pairs)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            return n_pairs * N / dt, [d.clone() for d in dst], c_l.colorlut_kernel_name()

        for rep in range(2):
            fps, ref, name = measure(one_stream)
            print("N=%d  one stream               %8.0f frames/s  (%s)" % (N, fps, name), flush=True)
            for ahead in (1, 2):
                fps, out, name = measure(lambda n: two_streams(n, ahead))
                same = all(bool((a == b).all()) for a, b in zip(ref, out))
                print("N=%d  two streams, ahead %d     %8.0f frames/s  (%s)  outputs identical: %s" % (N, ahead, fps, name, same), flush=True)
        del work, dst, master
        torch.cuda.empty_cache()
    c_h.close(); c_l.close()


if __name__ == "__main__":
    main()
