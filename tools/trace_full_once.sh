#!/bin/bash
# reproduces the "all legs" kernel trace of tools/collect_profiles.sh alone, keeping the log (round 5: it segfaulted once under rocprofv3)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/trace_full_once
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o t -- $PY $R/bench.py --no-live-pmc > "$OUT/trace_full.log" 2>&1
echo "rc=$?"
tail -30 "$OUT/trace_full.log"
f=$(find "$OUT/t" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" "$OUT/kernel_stats_full.csv" && head -30 "$OUT/kernel_stats_full.csv"
rm -rf "$OUT/t"
