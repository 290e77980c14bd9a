#!/bin/bash
# brick kernel: re-tune the work-sharing knobs on the chain's content (post-hsvfilter) after the round-3 VALU / LDS changes
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
run() { tools/exp_brick_build.sh "$2"; echo "=== $1 [$2]" >> $O/brick.log
  PREHSV=1 AMPS=0,4 SETS=32 VARIANTS=7 PRIO=${3:-3} TPR=${4:-0} timeout 300 python tools/bench_brick.py 2>&1 | tail -2 >> $O/brick.log; }
run base ""
run chunk1 "-DBRICK_CHUNK32=1"
run chunk4 "-DBRICK_CHUNK32=4"
run sb1 "-DBRICK_SB32=1"
run sb4 "-DBRICK_SB32=4"
run prio0 "" 0
run prio1 "" 1
run prio2 "" 2
run tpr16 "" 3 16
run tpr64 "" 3 64
run base2 ""
cat $O/brick.log
