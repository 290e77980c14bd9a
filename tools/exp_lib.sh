#!/bin/bash
# usage: tools/exp_lib.sh NAME "-DTAG_EXP=2 ..." [file.hip]  -- an experimental libmi355fx built HERE (cross-compiled) beside the real one:
# gst-plugins-rs_amd/exp/libmi355fx_NAME.so = the current objects with ONE file (default csrc/colorlut_window.hip) recompiled under
# extra flags. Travels to the GPU box with gpurun (*.so is git-ignored, not gpurun-ignored); pick it with MI355FX_LIB=...
set -e
cd "$(dirname "$0")/../gst-plugins-rs_amd"
name=$1; flags=$2; file=${3:-csrc/colorlut_window.hip}
mkdir -p exp
make -s all 2>&1 | grep -i "error" || true
obj=exp/$(basename "${file%.hip}")_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $flags -c "$file" -o "$obj"
others=$(ls csrc/*.o | grep -v "csrc/$(basename "${file%.hip}").o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp/libmi355fx_$name.so $others "$obj"
rm -f "$obj"
echo "built gst-plugins-rs_amd/exp/libmi355fx_$name.so"
