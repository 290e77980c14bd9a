#!/bin/bash
# first GPU call of round 3: hsv parity subset, A/B timings, memory-policy probe
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hsvfilter or hsvdetect" > gpurun_out/r03a/pytest_hsv.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03a/pytest_hsv.log
tail -5 gpurun_out/r03a/pytest_hsv.log
timeout 600 python tools/r03_hsv_ab.py > gpurun_out/r03a/hsv_ab.log 2>&1; tail -60 gpurun_out/r03a/hsv_ab.log
timeout 600 tools/hsv_mem_probe > gpurun_out/r03a/mem_probe.log 2>&1; tail -5 gpurun_out/r03a/mem_probe.log
