#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
