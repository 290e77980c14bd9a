#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
timeout 300 python bench.py --config 5 --steps 30 --warmup 3 > $O/config5.json 2>$O/config5.err; cat $O/config5.json | cut -c1-600
timeout 300 python tools/bench_dssim.py > $O/dssim.log 2>&1; tail -12 $O/dssim.log
timeout 200 python tools/r03_hsv_ab.py 2>&1 | grep "RGB 8x4K\|BGR 8x4K" 
