#!/bin/bash
# run ON THE GPU BOX: parity tests + bench of every experimental Dssim geometry build present
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for lib in gst-plugins-rs_amd/libmi355fx.so gst-plugins-rs_amd/libmi355fx_*x*.so; do
  echo "== $lib"
  MI355FX_LIB=$R/$lib timeout 600 python -m pytest tests/test_gpu_dssim.py -x -q 2>&1 | tail -1
  MI355FX_LIB=$R/$lib timeout 300 python tools/bench_dssim.py 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,4) for k,v in d.items() if k.endswith('_ms') or k.startswith('comparisons')})"
done
