#!/bin/bash
# brick kernel experiments: clamp modifier baseline, axis tables through the vector memory path
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
run() { # name, flags
  tools/exp_brick_build.sh "$2"
  echo "=== $1 [$2]" >> $O/brick.log
  AMPS=0,4,8 SETS=32,64 VARIANTS=7 timeout 300 python tools/bench_brick.py >> $O/brick.log 2>&1
  PREHSV=1 AMPS=0,4 SETS=32 VARIANTS=7 timeout 300 python tools/bench_brick.py >> $O/brick.log 2>&1
}
run base ""
run axis_all "-DBRICK_AXIS_GLOBAL=7"
run axis_z "-DBRICK_AXIS_GLOBAL=4"
run axis_yz "-DBRICK_AXIS_GLOBAL=6"
run base_again ""
cat $O/brick.log
