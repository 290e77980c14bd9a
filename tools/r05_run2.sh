#!/bin/bash
mkdir -p gpurun_out/r05
out=gpurun_out/r05/tag_exp_1.txt
: > $out
for lib in "" copy hit; do
  if [ -n "$lib" ]; then export MI355FX_LIB=$PWD/gst-plugins-rs_amd/exp/libmi355fx_$lib.so; else unset MI355FX_LIB; fi
  echo "== lib ${lib:-base}" >> $out
  VARIANTS=${VARIANTS:-8:1,8:0} timeout 300 python tools/window_probe.py 0 4 >> $out 2>&1
done
cat $out
