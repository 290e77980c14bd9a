"""CPU simulation of the block-shared LDS window cache of the memoised-table kernel (csrc/colorlut_window.hip): which share
of a frame's pixels find their table brick in the block's LDS, for candidate geometries, on the bench's frames.

A brick is a 4x4x4 colour cube (64 table entries = 256 B); the cache holds SETS x WAYS of them, the set is the brick's
position modulo a box of bricks, one brick per set is installed per step (a step = TILE_W x TILE_H pixels = what one block
looks up between two barriers); a block walks down its strip.  Run here (numpy + the C oracle for hsvfilter), no GPU.

  python tools/window_cache_sim.py [amp ...]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
from mi355fx import synth
from oracle import oracle

W, H = 3840, 2160


def frame(amp, hsv, seed=7):
    f = synth.smooth_frame(W, H, seed=seed).reshape(H, W, 4).astype(np.int16)
    if amp:
        rng = np.random.default_rng(2)
        f[..., :3] += rng.integers(-amp, amp + 1, size=f[..., :3].shape, dtype=np.int16)
    f = np.clip(f, 0, 255).astype(np.uint8)
    if hsv:
        buf = np.ascontiguousarray(f.reshape(-1))
        oracle.hsvfilter(buf, W, W * 4, 4, 0, False, synth.HSV_SETTINGS["hue90"], nthreads=8)
        f = buf.reshape(H, W, 4)
    return f


def simulate(f, box, ways, tile_w, tile_h, run_steps, fills_per_set=1):
    """box = (bx, by, bz) bricks per axis in one way. Returns (miss share of pixels, fills per step)."""
    bx, by, bz = box
    r, g, b = (f[..., c].astype(np.int32) >> 2 for c in range(3))  # brick coordinates 0..63
    sets = (r % bx) + bx * ((g % by) + by * (b % bz))
    tags = (r // bx) + 64 * ((g // by) + 64 * (b // bz))
    n_sets = bx * by * bz
    miss_px = 0
    fills = 0
    steps = 0
    n_strips = W // tile_w
    steps_per_strip = H // tile_h
    for sx in range(n_strips):
        cache = np.full((n_sets, ways), -1, np.int64)
        fifo = np.zeros(n_sets, np.int64)
        for sy in range(steps_per_strip):
            if sy % run_steps == 0:
                cache[:] = -1  # another block starts here on a cold cache
            s = sets[sy * tile_h:(sy + 1) * tile_h, sx * tile_w:(sx + 1) * tile_w].ravel()
            t = tags[sy * tile_h:(sy + 1) * tile_h, sx * tile_w:(sx + 1) * tile_w].ravel()
            hit = (cache[s] == t[:, None]).any(axis=1)
            miss_px += int((~hit).sum())
            steps += 1
            ms, mt = s[~hit], t[~hit]
            for _ in range(fills_per_set):
                if ms.size == 0:
                    break
                # one request per set survives (the last writer's); install it in the FIFO victim
                order = np.arange(ms.size)
                last = np.zeros(n_sets, np.int64) - 1
                last[ms] = order
                win = last[last >= 0]
                ws, wt = ms[win], mt[win]
                cache[ws, fifo[ws]] = wt
                fifo[ws] = (fifo[ws] + 1) % ways
                fills += ws.size
                keep = ~(cache[ms] == mt[:, None]).any(axis=1)
                ms, mt = ms[keep], mt[keep]
    return miss_px / (n_strips * steps_per_strip * tile_w * tile_h), fills / steps


def main():
    amps = [int(a) for a in sys.argv[1:]] or [0, 4, 8]
    geoms = [((8, 8, 8), 1), ((8, 8, 4), 2), ((8, 4, 4), 4)]
    tiles = [(256, 32), (512, 16), (128, 64)]
    for hsv in (True, False):
        for amp in amps:
            f = frame(amp, hsv)
            for box, ways in geoms:
                for tw, th in tiles:
                    m, fl = simulate(f, box, ways, tw, th, run_steps=32)
                    print("hsv=%d amp=%-3d box=%s ways=%d tile=%dx%d: miss %.2f %% of pixels, %.1f brick fills per step" %
                          (hsv, amp, box, ways, tw, th, 100 * m, fl), flush=True)


if __name__ == "__main__":
    main()
