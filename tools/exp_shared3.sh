#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_shared_brick.py tests/test_gpu_brick.py tests/test_gpu_parity.py tests/test_gpu_zz_timing.py -x -q -m gpu 2>&1 | tail -12 | tee gpurun_out/shared3_tests.log
timeout 900 python bench.py > gpurun_out/shared3_bench.json 2> gpurun_out/shared3_bench.err; tail -3 gpurun_out/shared3_bench.err
python tools/bench_summary.py gpurun_out/shared3_bench.json
