#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hrtf.py tests/test_gpu_sofa.py -x -q -m gpu > $O/pytest_hrtf.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hrtf.log
tail -25 $O/pytest_hrtf.log
for m in 1 2; do for taps in 64 128 256 512 1024 2048; do timeout 120 python tools/bench_hrtf.py --no-cpu --method $m --taps $taps 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config'], 'ms/block %.4f' % d['device_ms_per_block'], 'RT %.0f' % d['realtime_factor'])"; done; done 2>&1 | tee $O/hrtf_methods.log
timeout 300 python tools/bench_sofa.py 2>&1 | tee $O/sofa.log
