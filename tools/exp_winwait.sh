#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for wv in 0 1 0 1; do
  bash tools/exp_window_build.sh "-DWIN_WAIT=$wv"
  echo "== WIN_WAIT $wv"
  MODE=lutonly AMPS=0,4,8,16 CONFIGS=8:0 timeout 900 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tail -4
  MODE=chain AMPS=4,8 CONFIGS=8:0 timeout 900 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tail -2
done
bash tools/exp_window_build.sh "-DWIN_WAIT=1"
python -m pytest tests/test_gpu_window.py -q -m gpu -x 2>&1 | tail -2
