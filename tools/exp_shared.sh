#!/bin/bash
# one-shot: parity of the block-shared brick cache, then the A/B against the per-wave geometries on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_shared_brick.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/shared_tests.log
cat gpurun_out/shared_tests.log
VARIANTS=7,3 SETS=32,64,512 AMPS=0,2,4,8,16,32,-1 timeout 900 python tools/bench_brick.py 2>&1 | tee gpurun_out/shared_bench.log
PREHSV=1 VARIANTS=7 SETS=32,64,512 AMPS=0,4,8 timeout 900 python tools/bench_brick.py 2>&1 | tee gpurun_out/shared_bench_prehsv.log
bash tools/pmc_shared.sh 2>&1 | tee gpurun_out/shared_pmc.log
