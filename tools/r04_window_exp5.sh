#!/bin/bash
# round 4, GPU call: wave-local installs with the install loads and the missed lanes' gathers in one trip
out=gpurun_out/r04f; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_window.py -x -q 2>&1 | tail -15 | tee $out/win_tests.txt
for flags in "" "-DWIN_FILLS=4" "-DWIN_FILLS=12" "-DWIN_EXP=2" "-DWIN_EXP=1"; do
  tools/exp_window_build.sh "$flags"
  echo "== flags: $flags" | tee -a $out/exp5.txt
  VARIANTS=8 timeout 300 python tools/window_probe.py 0 4 8 16 2>&1 | grep -v "^fused.*variant [^8]" | tee -a $out/exp5.txt
done
tools/exp_window_build.sh ""
