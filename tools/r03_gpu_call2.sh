#!/bin/bash
# second GPU call of round 3: hsv parity, nt A/B inside the chain (bench.py), probe of the launch boundary
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "hsvfilter or hsvdetect" > $O/pytest_hsv.log 2>&1; echo "pytest rc=$?" >> $O/pytest_hsv.log
tail -5 $O/pytest_hsv.log
timeout 600 python tools/r03_hsv_ab.py > $O/hsv_ab.log 2>&1; tail -70 $O/hsv_ab.log
for nt in 1 0 1 0; do
  timeout 600 python bench.py --no-extra --no-cpu-baseline --ctx-flag 12=$nt > $O/bench_nt${nt}_$RANDOM.json 2>$O/bench_err.log
done
grep -h -o '"value": [0-9.]*\|"ctx_flags": [^]]*\]\|"hsvfilter_ms_per_launch": [0-9.]*\|"colorlut_ms_per_launch": [0-9.]*' $O/bench_nt*.json
