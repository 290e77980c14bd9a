#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_window.py -x -q > gpurun_out/r05/pytest_window_v2.txt 2>&1
tail -5 gpurun_out/r05/pytest_window_v2.txt
out=gpurun_out/r05/tag_exp_2.txt
: > $out
for lib in ""; do
  if [ -n "$lib" ]; then export MI355FX_LIB=$PWD/gst-plugins-rs_amd/exp/libmi355fx_$lib.so; else unset MI355FX_LIB; fi
  echo "== lib ${lib:-base}" >> $out
  VARIANTS=${VARIANTS:-5,8:0,8:2,8:3} timeout 300 python tools/window_probe.py 0 4 8 16 >> $out 2>&1
done
cat $out
