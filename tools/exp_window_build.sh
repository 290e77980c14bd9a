#!/bin/bash
# usage: tools/exp_window_build.sh "-DWIN_DEPTH=2 -DWIN_FILLS=8" -- rebuilds the LDS-cached table kernel with overrides (experiments on the GPU box)
cd "$(dirname "$0")/../gst-plugins-rs_amd" || exit 1
touch csrc/colorlut_window.hip
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $1" all 2>&1 | grep -i error
