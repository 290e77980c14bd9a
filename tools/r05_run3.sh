#!/bin/bash
mkdir -p gpurun_out/r05
./tools/walk_bench > gpurun_out/r05/walk_bench_3.txt 2>&1
MI355_TEST_WINDOW_ORDER=0 timeout 900 python -m pytest tests/test_gpu_window.py -x -q > gpurun_out/r05/pytest_window_v2.txt 2>&1
tail -5 gpurun_out/r05/pytest_window_v2.txt
VARIANTS=5,8:0,8:1:0,8:1:1 timeout 600 python tools/window_probe.py 0 4 8 > gpurun_out/r05/window_probe_v2.txt 2>&1
cat gpurun_out/r05/walk_bench_3.txt gpurun_out/r05/window_probe_v2.txt
