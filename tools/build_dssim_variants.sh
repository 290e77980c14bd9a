#!/bin/bash
# tools/build_dssim_variants.sh — experimental builds of the library with other Dssim tile geometries (development tool):
# gst-plugins-rs_amd/libmi355fx_<TW>x<TH>_<NT>.so, loaded with MI355FX_LIB=<path>
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/gst-plugins-rs_amd
make -s libmi355fx.so
OBJS=$(ls csrc/*.o | grep -v dssim_kernels)
for g in "$@"; do   # TWxTHxNT
  IFS=x read TW TH NT <<< "$g"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -DDSSIM_TW=$TW -DDSSIM_TH=$TH -DDSSIM_NT=$NT -c csrc/dssim_kernels.hip -o /tmp/dssim_$g.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmi355fx_$g.so $OBJS /tmp/dssim_$g.o
  echo built libmi355fx_$g.so
done
