#!/bin/bash
# tools/collect_profiles.sh <tag> — run ON THE GPU BOX (via gpurun). Collects, for the default bench.py
# workload: (1) rocprofv3 --kernel-trace --stats, (2) FETCH_SIZE and (3) WRITE_SIZE in separate --pmc
# passes (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit one pass; MI355X_MICROARCH.md §PMC slots).
# Outputs land in gpurun_out/profiles_<tag>/ ; tools/summarize_profiles.py condenses them into profiles/.
set -u
TAG=${1:-r01}
EXTRA=${2:-}   # extra bench.py arguments, e.g. "--lut-variant 6"
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# the interpreter itself, not a PATH name or a wrapper, goes behind `--`: the profiler has initialised the GPU before the program
# starts, and a launcher that re-executes is the exec this pool forbids (bench.py resolves it the same way for its child runs)
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
BENCH="$PY $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra $EXTRA"
# kernel traces: (1) the bench line's own command (all legs: the table kernel's average then mixes the two-launch leg,
# whose input is on-die, with the fused leg, whose input comes from HBM); (2) the same K / W with only the main leg, whose
# per-kernel averages are the ones roofline.avg_launch_ms must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_full" -o t -- $PY $R/bench.py --no-live-pmc $EXTRA > "$OUT/trace_full.log" 2>&1; echo "trace_full rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- $PY $R/bench.py --no-cpu-baseline --no-extra $EXTRA > "$OUT/trace.log" 2>&1; echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o f -- $BENCH > "$OUT/pmc_fetch.log" 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o w -- $BENCH > "$OUT/pmc_write.log" 2>&1; echo "write rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_tcc" -o c -- $BENCH > "$OUT/pmc_tcc.log" 2>&1; echo "tcc rc=$?"
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d "$OUT/pmc_tcp" -o p -- $BENCH > "$OUT/pmc_tcp.log" 2>&1; echo "tcp rc=$?"
grep -h '^{' "$OUT/trace.log" | tail -1 > "$OUT/bench_under_trace.json"
$PY $R/bench.py $EXTRA > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"
