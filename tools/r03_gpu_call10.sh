#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
timeout 600 python tools/bench_rgba64.py > $O/rgba64.log 2>&1; cat $O/rgba64.log
AMPS=0,4 SETS=32 VARIANTS=7 timeout 300 python tools/bench_brick.py 2>&1 | tail -3
