#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loudnorm.py -x -q -m gpu > $O/pytest_ln.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ln.log
tail -25 $O/pytest_ln.log
timeout 600 python tools/bench_loudnorm_batch.py > $O/ln_batch.log 2>&1; cat $O/ln_batch.log | tail -12
