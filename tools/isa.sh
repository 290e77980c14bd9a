#!/bin/bash
# usage: tools/isa.sh csrc/file.hip [extra flags] -- device ISA of one kernel file to /tmp/<name>.s plus its resource lines
cd /root/repo/gst-plugins-rs_amd
f=$1; shift
out=/tmp/$(basename "${f%.hip}").s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -S --cuda-device-only "$@" "$f" -o "$out" 2>&1 | grep error | head
grep -n "\.vgpr_count\|\.vgpr_spill\|\.name:" "$out" | paste - - - | awk '{print $3, $5, $7}'
