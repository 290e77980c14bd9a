#!/bin/bash
# round 4, GPU call: installs first, second look, gathers only for what is left; baselines on the same box
out=gpurun_out/r04g; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -15 | tee $out/win_tests.txt
for flags in "" "-DWIN_FILLS=4" "-DWIN_FILLS=16" "-DWIN_EXP=2" "-DWIN_EXP=1"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp6.txt
  VARIANTS=9,8,6 timeout 300 python tools/window_probe.py 0 4 8 16 2>&1 | tee -a $out/exp6.txt
done
tools/exp_window_build.sh ""
