#!/bin/bash
# tools/r06_miss_probe.sh — run ON THE GPU BOX: the miss path of colorlut_window_kernel in variants (tools/exp_lib.sh NAME FLAGS), same box, same call:
# ms per 8 x 4K launch over content noise (tools/window_probe.py), the cache's statistics, and the window tests under every variant
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_miss; rm -rf $O; mkdir -p $O
VAR=${VARS:-"ow ow2 ow3 ml2"}
for rep in 1 2; do
  echo "== default library, pass $rep"; VARIANTS=8:1,5 python3 tools/window_probe.py 0 4 8 16 2>&1 | grep -v "^input\|^fused"
  for v in $VAR; do
    echo "== $v, pass $rep"; MI355FX_LIB=$R/gst-plugins-rs_amd/exp/libmi355fx_$v.so VARIANTS=8:1 python3 tools/window_probe.py 0 4 8 16 2>&1 | grep -v "^input\|^fused"
  done
done > $O/probe.txt 2>&1
cat $O/probe.txt
for v in $VAR; do
  echo "== tests under $v"; MI355FX_LIB=$R/gst-plugins-rs_amd/exp/libmi355fx_$v.so python3 -m pytest tests/test_gpu_window.py -x -q -m gpu 2>&1 | tail -2
done | tee -a $O/probe.txt
