#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for sl in 1 3 5 8; do
  bash tools/exp_brick_build.sh "-DSH_SLACK=$sl"
  echo "== slack $sl"
  MODE=lutsame AMPS=0,4,8,16 CONFIGS=7:512 timeout 900 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee gpurun_out/shared_slack.log
