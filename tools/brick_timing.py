"""Reads the per-run records of a -DBRICK_TIMING build (BRICK_TIMING_FILE=...) and prints how evenly the waves finish:
lifetimes, finish times relative to the first start, and what a slow run has that a fast one does not (miss steps)."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
n_runs, tpr, n_strips, zn = [int(v) for v in raw[:4]]
r = raw[4:4 + 4 * n_runs].reshape(-1, 4)
start, end = r[:, 0].astype(np.float64), r[:, 1].astype(np.float64)
miss = (r[:, 2] & np.uint64(0xffffffff)).astype(np.float64); slow = (r[:, 2] >> np.uint64(32)).astype(np.float64)
ok = end > 0
start, end, miss, slow = start[ok], end[ok], miss[ok], slow[ok]
t0 = start.min()
life = end - start
total = end.max() - t0
print("waves %d tpr %d strips %d zn %d; kernel span %.0f ticks" % (n_runs, tpr, n_strips, zn, total))
q = [0, 10, 50, 90, 99, 100]
print("start  pct", q, np.percentile(start - t0, q).round(0) / total)
print("finish pct", q, (np.percentile(end - t0, q) / total).round(3))
print("life   pct", q, (np.percentile(life, q) / total).round(3))
print("miss steps per run pct", q, np.percentile(miss, q), " slow", np.percentile(slow, q))
tiles = (r[:, 3] & np.uint64(0xffffffff)).astype(np.float64)[ok]; stolen = (r[:, 3] >> np.uint64(32)).astype(np.float64)[ok]
print("tiles per wave pct", q, np.percentile(tiles, q), " of them stolen", np.percentile(stolen, q), " total stolen %.1f %%" % (100 * stolen.sum() / max(tiles.sum(), 1)))
c = np.corrcoef(life, miss)[0, 1]
A = np.vstack([np.ones_like(miss), miss, slow]).T
coef, *_ = np.linalg.lstsq(A, life, rcond=None)
print("corr(life, miss) %.3f; life ~ %.0f + %.1f x miss + %.1f x slow ticks" % (c, *coef))
print("mean life / span %.3f  (1.0 = every wave busy for the whole kernel)" % (life.mean() / total))
