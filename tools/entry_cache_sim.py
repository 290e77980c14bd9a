"""CPU simulation of an ENTRY cache for the memoised-table colorlut kernel: a direct-mapped LDS table of {colour, value} pairs (no
bricks, no locks: a miss reads the table and overwrites its entry), walked like colorlut_window_kernel (256-pixel strips, 32 rows
per step, a block walks down its strip). Which share of the pixels misses, on the bench's frames + noise? Two bounds: installs
visible after the whole step (all sixteen waves look up at once) and after every wave's 512 pixels (waves one after the other).
  python tools/entry_cache_sim.py [amp ...]     (numpy + the C oracle for hsvfilter; no GPU)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "gst-plugins-rs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from window_cache_sim import frame, W, H


def simulate(f, log2n, tile_w=256, tile_h=32, run_steps=34, per_wave=False):
    r, g, b = (f[..., c].astype(np.int64) for c in range(3))
    col = r | (g << 8) | (b << 16)
    # index: the low bits of every channel (a compact cloud of colours never collides with itself)
    nb = {13: (5, 4, 4), 14: (5, 5, 4), 15: (5, 5, 5), 16: (6, 5, 5)}[log2n]
    idx = (r & ((1 << nb[0]) - 1)) | ((g & ((1 << nb[1]) - 1)) << nb[0]) | ((b & ((1 << nb[2]) - 1)) << (nb[0] + nb[1]))
    miss = 0
    total = 0
    for sx in range(W // tile_w):
        cache = np.full(1 << log2n, -1, np.int64)
        for sy in range((H + tile_h - 1) // tile_h):
            if sy % run_steps == 0:
                cache[:] = -1   # another block starts here on a cold cache
            ii = idx[sy * tile_h:(sy + 1) * tile_h, sx * tile_w:(sx + 1) * tile_w]
            cc = col[sy * tile_h:(sy + 1) * tile_h, sx * tile_w:(sx + 1) * tile_w]
            groups = [(ii[k:k + 2].ravel(), cc[k:k + 2].ravel()) for k in range(0, ii.shape[0], 2)] if per_wave else [(ii.ravel(), cc.ravel())]
            for i, c in groups:
                hit = cache[i] == c
                miss += int((~hit).sum())
                total += i.size
                cache[i[~hit]] = c[~hit]
    return miss / total


def main():
    amps = [int(a) for a in sys.argv[1:]] or [0, 2, 4, 8]
    for hsv in (True, False):
        for amp in amps:
            f = frame(amp, hsv)
            row = []
            for log2n in (13, 14, 15):
                row.append("2^%d: %.1f %% / %.1f %%" % (log2n, 100 * simulate(f, log2n), 100 * simulate(f, log2n, per_wave=True)))
            print("%-9s amp %-2d  pixels that miss (step-synchronous / wave after wave)  " % ("post-hsv" if hsv else "pristine", amp) + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
