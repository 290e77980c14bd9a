#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 600 python tools/r03_hsv_ab.py > $O/hsv_ab.log 2>&1; tail -40 $O/hsv_ab.log
