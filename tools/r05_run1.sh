#!/bin/bash
# round 5, first GPU call: pair-atomicity probe, window tests for both kernels, A/B probe
mkdir -p gpurun_out/r05
./tools/lds_pair_probe > gpurun_out/r05/lds_pair_probe.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_window.py -x -q > gpurun_out/r05/pytest_window.txt 2>&1
tail -5 gpurun_out/r05/pytest_window.txt
timeout 600 python tools/window_probe.py 0 4 8 16 > gpurun_out/r05/window_probe_1.txt 2>&1
cat gpurun_out/r05/lds_pair_probe.txt gpurun_out/r05/window_probe_1.txt
