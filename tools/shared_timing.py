"""Debug (BRICK_TIMING build): per-wave start / end times of one colorlut3d_shared_kernel launch -> spread of the blocks' finish times.
   BRICK_TIMING_FILE=/tmp/t.bin python tools/run_colorlut_once.py 7 3 <amp> 512 ; python tools/shared_timing.py /tmp/t.bin"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64)
nw = int(a[0]); tr = a[4:4 + 4 * nw].reshape(nw, 4)
t0 = tr[:, 0].astype(np.int64); t1 = tr[:, 1].astype(np.int64)
base = t0.min()
us = lambda t: (t - base) / 100.0   # s_memrealtime: 100 MHz
blocks = nw // 16
end_b = us(t1).reshape(blocks, 16).max(axis=1); start_b = us(t0).reshape(blocks, 16).min(axis=1)
miss = (tr[:, 2] & 0xffffffff).astype(np.int64).reshape(blocks, 16).sum(axis=1)
slow = (tr[:, 2] >> 32).astype(np.int64).reshape(blocks, 16).sum(axis=1)
steps = ((tr[:, 3] >> 32).astype(np.int64) - (tr[:, 3] & 0xffffffff).astype(np.int64)).reshape(blocks, 16)[:, 0]
dur = end_b - start_b
print("blocks %d  steps per block %d..%d" % (blocks, steps.min(), steps.max()))
print("block start  us: min %.1f  median %.1f  max %.1f" % (start_b.min(), np.median(start_b), start_b.max()))
print("block end    us: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f" % (end_b.min(), np.percentile(end_b, 10), np.median(end_b), np.percentile(end_b, 90), end_b.max()))
print("block duration us: min %.1f  median %.1f  mean %.1f  max %.1f" % (dur.min(), np.median(dur), dur.mean(), dur.max()))
wave_end = us(t1).reshape(blocks, 16)
print("within a block, last wave end - first wave end: median %.1f us  max %.1f us" % (np.median(wave_end.max(1) - wave_end.min(1)), (wave_end.max(1) - wave_end.min(1)).max()))
c = np.corrcoef(dur, miss)[0, 1]
print("miss half-steps per block: min %d median %d max %d; correlation with duration %.2f; slow: median %d max %d" % (miss.min(), np.median(miss), miss.max(), c, np.median(slow), slow.max()))
order = np.argsort(dur)
for i in list(order[:3]) + list(order[-5:]):
    print("  block %3d  dur %.1f us  start %.1f  miss %d  slow %d" % (i, dur[i], start_b[i], miss[i], slow[i]))
# per strip: mean block duration (blocks are numbered along the strips)
sps = int(a[1]); ns = int(a[2])
firsts = (tr[:, 3] & 0xffffffff).astype(np.int64).reshape(blocks, 16)[:, 0]
strip_of = firsts // sps
print("strip: mean duration us / mean miss   " + "  ".join("%d: %.0f/%d" % (s_, dur[strip_of == s_].mean(), miss[strip_of == s_].mean()) for s_ in range(ns) if (strip_of == s_).any()))
