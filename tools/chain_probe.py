"""Interpolating colorlut kernels IN the chain (hsvfilter in place, then colorlut, on batches streamed from HBM - the brick table
is then cold in L2 at every colorlut launch, unlike in tools/bench_brick.py's back-to-back colorlut launches), per noise
amplitude and kernel: pinned per-wave caches (32 / 64 sets), the block-shared cache (512) and the content watch (variant 6).
Run on the GPU box: python tools/chain_probe.py            (AMPS=0,4,8  CONFIGS=7:32,7:64,7:512,6:0  N=8)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, ROOT)
import torch
import mi355fx
from mi355fx import synth
from mi355fx.cube import parse_cube
import bench

W, H = bench.W, bench.H
N = int(os.environ.get("N", "8"))


def main():
    dev = torch.device("cuda:0")
    ctx = mi355fx.Context(0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    lut = parse_cube(synth.cube_text_3d(int(os.environ.get("LUT_SIZE", "33"))))
    settings = synth.HSV_SETTINGS["hue90"]
    pitch = W * H * 4
    configs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "7:32,7:64,7:512,6:0").split(",")]
    print("%-6s %s" % ("amp", "  ".join("v%d/sets %-3d lut ms [hsv ms] kernel" % c for c in configs)))
    for amp in [int(a) for a in os.environ.get("AMPS", "0,4,8,16").split(",")]:
        pool = bench.SourcePool(torch, synth, dev, N, "smooth+%d" % amp if amp else "smooth")
        pristine = [pool.new(k) for k in range(6)]
        if os.environ.get("FRAMES") == "seeds":   # tools/bench_brick.py's frames: eight different seeds, no rotation
            rng = np.random.default_rng(2)
            for b in pristine:
                f = np.stack([synth.smooth_frame(W, H, seed=7 + i) for i in range(N)]).reshape(N, H, W, 4).astype(np.int16)
                if amp:
                    f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
                b.copy_(torch.from_numpy(np.clip(f, 0, 255).astype(np.uint8).reshape(N, H, W * 4)))
        if os.environ.get("FRAMES") == "norot":   # the pool's base frame, not rotated
            for b in pristine:
                for i in range(N):
                    b[i].copy_(pool.bases[i % 4])
        work = [torch.empty_like(pristine[0]) for _ in range(3)]
        dst = [torch.empty_like(pristine[0]) for _ in range(2)]
        cells = []
        for v, sets in configs:
            ctx.colorlut_load(lut.is3d, lut.size, lut.table, lut.domain_scale, lut.domain_offset)
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, v)
            ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
            evs, names = [], {}
            ctx.colorlut_brick_stats(reset=True)
            mode = os.environ.get("MODE", "chain")   # chain | copylut (pre-filtered input copied, then colorlut) | lutonly (pre-filtered input in place)
            if mode != "chain" and not getattr(pool, "filtered", False):
                for b in pristine:
                    ctx.hsvfilter_frames_device(b.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
                pool.filtered = True
            import time
            t_ramp = time.perf_counter()
            k_ramp = 0
            while time.perf_counter() - t_ramp < float(os.environ.get("RAMP", "0.4")):   # clocks up, watch settled: untimed
                for _ in range(8):
                    w_ = work[k_ramp % 3] if not mode.startswith("lut") else pristine[k_ramp % 6 if mode == "lutonly" else 0]
                    if not mode.startswith("lut"):
                        w_.copy_(pristine[k_ramp % 6])
                    if mode == "chain":
                        ctx.hsvfilter_frames_device(w_.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
                    ctx.colorlut_frames_device(w_.data_ptr(), pitch, W * 4, dst[k_ramp % 2].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
                    k_ramp += 1
                torch.cuda.synchronize()
            ctx.colorlut_brick_stats(reset=True)
            for k in range(int(os.environ.get("STEPS", "120"))):
                w_ = work[k % 3] if not mode.startswith("lut") else pristine[k % 6 if mode == "lutonly" else 0]
                if not mode.startswith("lut"):
                    w_.copy_(pristine[k % 6])
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                if mode == "chain":
                    ctx.hsvfilter_frames_device(w_.data_ptr(), N, pitch, W, H, W * 4, "RGBA", settings)
                e1.record()
                ctx.colorlut_frames_device(w_.data_ptr(), pitch, W * 4, dst[k % 2].data_ptr(), pitch, W * 4, N, W, H, "RGBA")
                e2.record()
                if k >= 40:
                    evs.append((e0, e1, e2))
                    nm = ctx.colorlut_kernel_name().replace("colorlut3d_", "")
                    names[nm] = names.get(nm, 0) + 1
            torch.cuda.synchronize()
            hs = sorted(a.elapsed_time(b) for a, b, _ in evs)
            ls = sorted(b.elapsed_time(c) for _, b, c in evs)
            st = ctx.colorlut_brick_stats(reset=True)
            tot = float(os.environ.get("STEPS", "120")) * N * W * H / 256
            cells.append("%.4f (min %.4f) [%.4f] miss %.3f slow %.3f %s" % (ls[len(ls) // 2], ls[0], hs[len(hs) // 2], st[0] / tot, st[1] / tot, ",".join("%s:%d" % kv for kv in sorted(names.items()))))
        print("%-6d %s" % (amp, "  ".join(cells)), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
