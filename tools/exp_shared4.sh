#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for f in pool seeds norot; do echo "== frames $f"; FRAMES=$f MODE=lutsame AMPS=0,8 CONFIGS=7:32,7:512 timeout 900 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/chain_probe.log; done
