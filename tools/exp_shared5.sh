#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_shared_brick.py -x -q -m gpu 2>&1 | tail -3
for f in pool seeds; do echo "== frames $f"; FRAMES=$f MODE=lutsame AMPS=0,4,8,16 CONFIGS=7:32,7:64,7:512 timeout 900 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/chain_probe.log; done
bash tools/exp_brick_build.sh "-DBRICK_TIMING"
for rot in 1; do
  echo "== rot '$rot'"
  ROT=$rot BRICK_TIMING_FILE=/tmp/t.bin python tools/run_colorlut_once.py 7 3 0 512 2>&1 | grep -v amdgpu.ids
  python tools/shared_timing.py /tmp/t.bin
done 2>&1 | tee gpurun_out/shared_timing.log
