"""Differential stress test on the GPU of the two kernels that cache bricks in LDS per CU with lock-free readers
(colorlut_window_kernel: table bricks, variant 8; colorlut3d_shared_kernel: LUT bricks, variant 7 + MI355_FLAG_BRICK_SETS 512):
random LUTs (3D sizes 2..65, non-unit domains), random geometry (widths that are / are not whole 256-pixel strips, heights that
are not whole 32-row steps, several frames per launch - always large enough for the kernels to be chosen), random content built to
be hard on a set-associative cache: patches of flat colour, gradients, noise of random amplitude, colours exactly 64 / 128 levels
apart side by side (same sets), in place and out of place. Each launch must equal the per-wave brick kernel's (variant 7, 32
sets), byte for byte; the kernel that served is checked by name. Device against device: hundreds of cases in a few minutes.
Run on the GPU box: python tools/stress_lds_caches.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
import mi355fx


def content(rng, rows, w):
    img = np.zeros((rows, w, 4), np.int16)
    kind = rng.integers(0, 5)
    yy, xx = np.mgrid[0:rows, 0:w]
    if kind == 0:      # gradient + bars an exact multiple of 64 levels apart
        step = int(rng.choice([64, 128, 192]))
        bars = (xx * int(rng.integers(2, 9)) // w) % 3
        for c in range(3):
            img[..., c] = (xx * int(rng.integers(0, 3)) + yy * int(rng.integers(0, 3))) * 255 // (rows + w) // 3 + ((bars + c) % 3 == 0) * step
    elif kind == 1:    # flat patches
        py, px = int(rng.integers(8, 200)), int(rng.integers(8, 300))
        pal = rng.integers(0, 256, (64, 3), dtype=np.int16)
        idx = ((yy // py) * 7 + (xx // px) * 13) % 64
        img[..., :3] = pal[idx]
    elif kind == 2:    # slow gradients in all channels
        for c in range(3):
            img[..., c] = (xx * int(rng.integers(1, 4)) + yy * int(rng.integers(1, 4)) + int(rng.integers(0, 256)) * 16) // 16 % 256
    elif kind == 3:    # two interleaved populations per row pair
        a, b = rng.integers(0, 256, 3, dtype=np.int16), rng.integers(0, 256, 3, dtype=np.int16)
        img[..., :3] = np.where(((yy // 2 + xx // 64) % 2 == 0)[..., None], a, b)
    else:              # uniform noise
        img[..., :3] = rng.integers(0, 256, (rows, w, 3), dtype=np.int16)
    amp = int(rng.choice([0, 0, 2, 5, 9, 20]))
    if amp:
        img[..., :3] += rng.integers(-amp, amp + 1, (rows, w, 3), dtype=np.int16)
    img[..., 3] = rng.integers(0, 256, (rows, w), dtype=np.int16)
    return np.clip(img, 0, 255).astype(np.uint8)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = mi355fx.Context(0)
    bad = 0
    served = {}
    for it in range(cases):
        size = int(rng.choice([2, 3, 9, 16, 17, 25, 32, 33, 34, 40, 65]))
        table = rng.uniform(-0.1, 1.1, (size ** 3, 4)).astype(np.float32); table[:, 3] = 1.0
        lo = rng.uniform(-0.2, 0.1, 3).astype(np.float32); hi = (lo + rng.uniform(0.6, 1.4, 3)).astype(np.float32)
        scale = (np.float32(1.0) / (hi - lo)).astype(np.float32); offset = (-lo * scale).astype(np.float32)
        if rng.integers(0, 3) == 0:
            scale[:] = 1.0; offset[:] = 0.0
        ctx.colorlut_load(True, size, table, scale, offset)
        w = int(rng.choice([256, 260, 512, 1000, 1280, 1920, 1924, 3840, 4096, 64, 8]))
        px_needed = int(rng.integers(7_000_000, 20_000_000))
        h = int(rng.choice([1080, 2160, 719, 33, 1000]))
        h = max(1, min(h, px_needed // w))
        n = max(1, -(-px_needed // (w * h)))
        if (n * h + 31) // 32 * ((w // 4 + 63) // 64) < 800 or n * h * w * 4 > (1 << 31) - 4096:
            continue
        frames = content(rng, n * h, w).reshape(-1)
        nb = frames.nbytes
        in_place = bool(rng.integers(0, 2))
        d_s, d_o = ctx.alloc(nb), ctx.alloc(nb)
        outs = {}
        for name, variant, sets in (("wave32", 7, 32), ("shared", 7, 512), ("window", 8, 0)):
            ctx.set_flag(mi355fx.FLAG_LUT_VARIANT, variant)
            ctx.set_flag(mi355fx.FLAG_BRICK_SETS, sets)
            ctx.h2d(d_s, frames)
            dst = d_s if in_place else d_o
            for rep in range(2 if name != "wave32" else 1):   # (the second launch of the table kernel runs on the finished table)
                if rep:
                    ctx.h2d(d_s, frames)
                ctx.colorlut_frames_device(d_s, w * h * 4, w * 4, dst, w * h * 4, w * 4, n, w, h, "RGBA")
            ctx.synchronize()
            k = ctx.colorlut_kernel_name()
            served[(name, k)] = served.get((name, k), 0) + 1
            o = np.empty_like(frames); ctx.d2h(o, dst); outs[name] = o
        ctx.free(d_s); ctx.free(d_o)
        ok = (outs["shared"] == outs["wave32"]).all() and (outs["window"] == outs["wave32"]).all()
        if not ok:
            bad += 1
            print("MISMATCH case", it, dict(size=size, w=w, h=h, n=n, in_place=in_place,
                                            shared=int((outs["shared"] != outs["wave32"]).sum()), window=int((outs["window"] != outs["wave32"]).sum())), flush=True)
    print("%d cases, %d mismatches; kernels served: %s" % (cases, bad, ", ".join("%s->%s x%d" % (a, b, c) for (a, b), c in sorted(served.items()))))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
