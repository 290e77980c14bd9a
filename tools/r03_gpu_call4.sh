#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
AMPS=0,4,8 SETS=32,64 VARIANTS=7 timeout 300 python tools/bench_brick.py > $O/brick.log 2>&1
PREHSV=1 AMPS=0,4 SETS=32 VARIANTS=7 timeout 300 python tools/bench_brick.py >> $O/brick.log 2>&1
cat $O/brick.log
timeout 900 python bench.py > $O/bench.json 2>$O/bench_err.log; tail -3 $O/bench_err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03d/bench.json"))
print("value", d["value"], "frac", d["roofline"]["frac"], d["roofline"]["kernel"])
print("kernels", {k:d["kernels"][k] for k in ("hsvfilter_ms_per_launch","colorlut_ms_per_launch","colorlut_kernels_served")})
for k in ("interpolating_kernel_only","fused_chain","other_content","concurrent_streams","cpu_baseline"):
    print(k, d.get(k))
PY
