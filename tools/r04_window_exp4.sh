#!/bin/bash
# round 4, GPU call: why did the hit path get slower with the fill wave?
out=gpurun_out/r04e; mkdir -p $out
for flags in "-DWIN_EXP=3" "-DWIN_EXP=2 -DWIN_SLEEP=127" "-DWIN_EXP=2 -DWIN_FILLPRIO=0" "-DWIN_EXP=2 -DWIN_DEPTH=1" "-DWIN_EXP=1"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp4.txt
  VARIANTS=8 timeout 300 python tools/window_probe.py 0 2>&1 | grep -v "^fused.*variant [^8]" | tee -a $out/exp4.txt
done
tools/exp_window_build.sh ""
