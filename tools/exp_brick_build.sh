#!/bin/bash
# usage: tools/exp_brick_build.sh "-DBRICK_W64=3 -DBRICK_BLOCKS64=3" -- rebuilds the brick kernel with overrides (experiments on the GPU box)
cd "$(dirname "$0")/../gst-plugins-rs_amd" || exit 1
touch csrc/colorlut_brick.hip
make -s HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $1" all 2>&1 | grep -i error
