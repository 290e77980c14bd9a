#!/usr/bin/env python3
"""tools/launch_gaps.py <dir with a rocprofv3 --kernel-trace csv> — what lies BETWEEN the chain's launches: for the hsvfilter /
colorlut kernels of the trace, in start order, the gap from one kernel's end to the next one's start (a dependent launch boundary
on one stream), next to the kernels' own durations. The bench line's ms_per_step minus its kernels' time is this, per launch.
Prints a small table (median / mean / p90 of the gaps over the last N launches = the timed region)."""
import csv, glob, os, sys


def main():
    d = sys.argv[1]
    n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not paths:
        print("no kernel_trace.csv under", d)
        return 1
    rows = []
    for row in csv.DictReader(open(paths[0])):
        name = row.get("Kernel_Name", "")
        if "mi355::" not in name:
            continue
        short = name.split("mi355::")[1].split("(")[0]
        if not (short.startswith("hsvfilter_flat_kernel") or short.startswith("colorlut_")):
            continue
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), short.split("<")[0]))
    rows.sort()
    rows = rows[-n_last:]
    gaps, by_pair, dur = [], {}, {}
    for (s0, e0, k0), (s1, e1, k1) in zip(rows, rows[1:]):
        g = (s1 - e0) / 1000.0
        if g > 200.0:      # a host-side pause between legs, not a launch boundary
            continue
        gaps.append(g)
        by_pair.setdefault(k0 + " -> " + k1, []).append(g)
    for s, e, k in rows:
        dur.setdefault(k, []).append((e - s) / 1000.0)

    def stats(v):
        v = sorted(v)
        return "n %4d  median %6.2f us  mean %6.2f us  p90 %6.2f us" % (len(v), v[len(v) // 2], sum(v) / len(v), v[int(len(v) * 0.9)])

    print("kernel durations over the last %d chain launches of the trace:" % len(rows))
    for k, v in sorted(dur.items()):
        print("  %-34s %s" % (k, stats(v)))
    print("gap between the end of a launch and the start of the next (same stream, dependent):")
    print("  %-34s %s" % ("all", stats(gaps)))
    for k, v in sorted(by_pair.items()):
        if len(v) >= 8:
            print("  %-58s %s" % (k, stats(v)))
    tot_k = sum(sum(v) for v in dur.values())
    print("share of the wall time between first start and last end that no chain kernel covers: %.1f %%" % (100.0 * (1.0 - tot_k / ((rows[-1][1] - rows[0][0]) / 1000.0))))
    return 0


if __name__ == "__main__":
    sys.exit(main())
