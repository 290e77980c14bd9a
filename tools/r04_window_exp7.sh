#!/bin/bash
# round 4, GPU call: wave-local installs (WIN_FW=0) against a fill wave + waiting pixel waves (WIN_FW=1), same box
out=gpurun_out/r04h; mkdir -p $out
for flags in "-DWIN_FW=1" "-DWIN_FW=0" "-DWIN_FW=1 -DWIN_EXP=1" "-DWIN_FW=0 -DWIN_EXP=1" "-DWIN_FW=1 -DWIN_EXP=2" "-DWIN_FW=0 -DWIN_EXP=2" "-DWIN_FW=1 -DWIN_FILLS=16" "-DWIN_FW=1 -DWIN_POLL=1" "-DWIN_FW=1 -DWIN_POLL=4 -DWIN_SLEEP=2"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp7.txt
  case "$flags" in *EXP*) amps="0";; *) amps="0 4 8 16";; esac
  VARIANTS=8 timeout 300 python tools/window_probe.py $amps 2>&1 | grep -v "^fused.*variant [^8]" | tee -a $out/exp7.txt
  case "$flags" in *EXP*) ;; *) timeout 600 python -m pytest tests/test_gpu_window.py -q 2>&1 | tail -3 | tee -a $out/exp7.txt;; esac
done
tools/exp_window_build.sh ""
