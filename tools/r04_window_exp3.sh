#!/bin/bash
# round 4, GPU call: the LDS-cached table kernel with a dedicated fill wave
out=gpurun_out/r04d; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -15 | tee $out/win_tests.txt
for flags in "" "-DWIN_EXP=2" "-DWIN_FILLS=16" "-DWIN_DEPTH=3"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp3.txt
  VARIANTS=8 timeout 300 python tools/window_probe.py 0 4 8 16 2>&1 | grep -v "^fused.*variant [^8]" | tee -a $out/exp3.txt
done
tools/exp_window_build.sh ""
