#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
VARIANTS=7,3 SETS=512 AMPS=8,12,16,20,24,28,32 timeout 900 python tools/bench_brick.py 2>&1 | tee gpurun_out/shared_bench_thresholds.log
PREHSV=1 VARIANTS=7,3 SETS=512 AMPS=8,12,16,20,24 timeout 900 python tools/bench_brick.py 2>&1 | tee -a gpurun_out/shared_bench_thresholds.log
for n in 1 2 4; do
echo "== $n frame(s) per launch"
N=$n VARIANTS=7,3 SETS=32,64,512 AMPS=0,4,8,16 timeout 900 python tools/bench_brick.py 2>&1 | tee -a gpurun_out/shared_bench_thresholds.log
done
