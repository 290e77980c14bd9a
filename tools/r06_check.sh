#!/bin/bash
# tools/r06_check.sh — run ON THE GPU BOX: the whole GPU suite, the LDS-DMA walk probe, config 5 (dssim / blockhash, with and without
# the dispatcher), the native-thread group bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_check; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -12 $O/pytest_gpu.txt
timeout 300 tools/walk_dma_bench > $O/walk_dma_bench.txt 2>&1; echo "walk rc=$?" >> $O/walk_dma_bench.txt
cat $O/walk_dma_bench.txt
for A in "" "--no-group" "--hash-algo blockhash" "--hash-algo blockhash --no-group"; do
  python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline $A 2>> $O/config5.err | grep '^{' | tail -1 | sed "s/^{/{\"args\": \"$A\", /" >> $O/config5_variants.jsonl
done
python3 - $O <<'PY'
import json, sys
for l in open(sys.argv[1] + "/config5_variants.jsonl"):
    x = json.loads(l); print("config5 [%s]" % x["args"], round(x["value"], 1), x.get("dispatcher"))
PY
timeout 900 tools/agroup_bench 32 > $O/agroup_bench_32.jsonl 2> $O/agroup_bench.err
timeout 600 tools/agroup_bench 8 > $O/agroup_bench_8.jsonl 2>> $O/agroup_bench.err
cat $O/agroup_bench_32.jsonl $O/agroup_bench_8.jsonl; tail -n 3 $O/*.err
