#!/bin/bash
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_window.py -x -q -k "bricktags" > gpurun_out/r05/pytest_window_tile.txt 2>&1
tail -4 gpurun_out/r05/pytest_window_tile.txt
out=gpurun_out/r05/tile_exp_2.txt
: > $out
for lib in "" d3; do
  if [ -n "$lib" ]; then export MI355FX_LIB=$PWD/gst-plugins-rs_amd/exp/libmi355fx_$lib.so; else unset MI355FX_LIB; fi
  echo "== lib ${lib:-base}" >> $out
  VARIANTS=${VARIANTS:-5,8:0:0,8:0:1} timeout 300 python tools/window_probe.py 0 4 >> $out 2>&1
done
cat $out
