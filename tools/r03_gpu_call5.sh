#!/bin/bash
# brick kernel: same-box sensitivity to VALU / LDS work per pixel, PIPE on/off
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
run() { tools/exp_brick_build.sh "$2"; echo "=== $1 [$2]" >> $O/brick.log
  AMPS=0,4 SETS=32 VARIANTS=7 timeout 300 python tools/bench_brick.py >> $O/brick.log 2>&1; }
run base ""
run valu4 "-DBRICK_DUMMY_VALU=4"
run valu8 "-DBRICK_DUMMY_VALU=8"
run lds1 "-DBRICK_DUMMY_LDS=1"
run lds2 "-DBRICK_DUMMY_LDS=2"
run nopipe "-DBRICK_PIPE=0"
run base2 ""
cat $O/brick.log
