#!/bin/bash
# round 4, GPU call: what bounds the LDS-cached table kernel on the pristine frames
out=gpurun_out/r04b; mkdir -p $out
for flags in "" "-DWIN_DEPTH=2" "-DWIN_EXP=1" "-DWIN_EXP=1 -DWIN_DEPTH=2" "-DWIN_EXP=2" "-DWIN_EXP=2 -DWIN_DEPTH=2" "-DWIN_FILLS=8" "-DWIN_FILLS=2"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp1.txt
  VARIANTS=8 timeout 300 python tools/window_probe.py 0 4 8 2>&1 | grep -v "^fused.*variant [^8]" | tee -a $out/exp1.txt
done
tools/exp_window_build.sh ""
timeout 900 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -15 | tee $out/win_tests.txt
