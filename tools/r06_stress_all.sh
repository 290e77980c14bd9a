#!/bin/bash
# tools/r06_stress_all.sh — run ON THE GPU BOX: every device-against-device stress tool on the final code of round 6
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_stress; rm -rf $O; mkdir -p $O
{
echo "== stress_auto.py 3000 61 (kernel choice: auto against pinned, random call sequences)"; timeout 600 python3 tools/stress_auto.py 3000 61 2>&1 | tail -3
echo "== stress_auto.py 3000 62"; timeout 600 python3 tools/stress_auto.py 3000 62 2>&1 | tail -3
echo "== stress_tables.py (memoised tables: shared registry, rebuilds, settings changes)"; timeout 600 python3 tools/stress_tables.py 2>&1 | tail -3
echo "== stress_lds_caches.py (LDS-cached kernels against the per-wave kernel on colliding content)"; timeout 900 python3 tools/stress_lds_caches.py 2>&1 | tail -3
echo "== stress_group.py 16 300 6 (video dispatcher, 16 threads)"; timeout 900 python3 tools/stress_group.py 16 300 6 2>&1 | tail -3
echo "== stress_dssim.py (fused Dssim pass against create + compare)"; timeout 900 python3 tools/stress_dssim.py 2>&1 | tail -3
echo "== stress_dispatch.py 30 (round-6 dispatchers)"; timeout 300 python3 tools/stress_dispatch.py 30 2>&1 | tail -3
} > $O/r06_stress_all.txt 2>&1
cat $O/r06_stress_all.txt
