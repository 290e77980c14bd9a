#!/bin/bash
# tools/batch_sweep.sh — run ON THE GPU BOX: bench.py over frames-per-launch (two-kernel chain, auto kernel choice)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for n in 1 2 4 6 8 12 16 32 64; do python3 $R/bench.py --batch $n --steps 100 --no-cpu-baseline --no-extra 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print($n, round(d['value']), round(k['hsvfilter_ms_per_launch'],4), round(k['colorlut_ms_per_launch'],4), k['colorlut_kernel'])"; done
