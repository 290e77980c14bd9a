#!/bin/bash
# tools/prof_dssim.sh — run ON THE GPU BOX: rocprofv3 kernel stats of tools/bench_dssim.py (per-kernel averages and the
# slowest launch = scale 0) into gpurun_out/prof_dssim/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dssim -o d -- python3 $R/tools/bench_dssim.py > $R/gpurun_out/prof_dssim.log 2>&1
python3 - "$R/gpurun_out/prof_dssim/d_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"].split("(")[0][-50:], r["Calls"], "avg %.1f us" % (float(r["AverageNs"]) / 1e3), "max %.1f us" % (float(r["MaxNs"]) / 1e3), r["Percentage"])
PY
