#!/usr/bin/env python3
"""tools/trace_big_gaps.py <dir with a rocprofv3 --kernel-trace csv> [n_last] — where the device idles inside the timed region:
every gap of more than 30 us between consecutive kernels (ALL kernels of the process, in start order) among the last n launches,
with its position and neighbours, and the sum of all gaps against the kernels' time."""
import csv, glob, os, sys

d = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 640
paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for row in csv.DictReader(open(paths[0])):
    name = row.get("Kernel_Name", "")
    short = name.split("mi355::")[1].split("(")[0].split("<")[0] if "mi355::" in name else name.split("(")[0][:40]
    rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
tot_gap = tot_k = 0.0
print("last %d launches span %.2f ms" % (len(rows), (rows[-1][1] - t0) / 1e6))
for i, ((s0, e0, k0), (s1, e1, k1)) in enumerate(zip(rows, rows[1:])):
    g = (s1 - e0) / 1e3
    tot_gap += max(g, 0.0)
    tot_k += (e0 - s0) / 1e3
    if g > 30.0:
        print("  after launch %4d (t = %8.2f ms)  gap %9.1f us   %s -> %s" % (i, (e0 - t0) / 1e6, g, k0, k1))
print("kernels %.2f ms, gaps %.2f ms" % (tot_k / 1e3, tot_gap / 1e3))
