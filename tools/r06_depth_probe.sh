#!/bin/bash
# tools/r06_depth_probe.sh — run ON THE GPU BOX: colorlut_window_kernel with two (default) and three steps of pixels in flight per lane
# (tools/exp_lib.sh depth3 "-DWIN_DEPTH=3"), same box, same call: from HBM over content noise, and the fused launch of the bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_depth; rm -rf $O; mkdir -p $O
for rep in 1 2; do
  echo "== depth 2 (default library), pass $rep"; VARIANTS=8:1 python3 tools/window_probe.py 0 4 8 2>&1 | grep -v "^input"
  echo "== depth 3, pass $rep"; MI355FX_LIB=$R/gst-plugins-rs_amd/exp/libmi355fx_depth3.so VARIANTS=8:1 python3 tools/window_probe.py 0 4 8 2>&1 | grep -v "^input"
done > $O/probe.txt 2>&1
cat $O/probe.txt
for lib in "" "$R/gst-plugins-rs_amd/exp/libmi355fx_depth3.so"; do
  MI355FX_LIB=$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-live-pmc --streams 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('lib [%s]: fused %.0f frames/s (%.4f ms per launch, %s), headline %.0f' % ('$lib'[-25:], d['fused_chain']['frames_per_s'], d['fused_chain']['ms_per_launch'], d['fused_chain'].get('kernels_served'), d['value']))"
done | tee -a $O/probe.txt
