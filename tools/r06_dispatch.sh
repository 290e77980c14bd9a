#!/bin/bash
# tools/r06_dispatch.sh — run ON THE GPU BOX: the round-6 dispatchers: their tests, config 5 through the compare queue, the audio group bench
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_dispatch; rm -rf $O; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_group_compare.py tests/test_gpu_agroup.py -q -m gpu > $O/pytest_new.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_new.txt
tail -30 $O/pytest_new.txt
for L in 1 2 4 8 16; do
  python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --compare-lanes $L 2>> $O/config5.err | grep '^{' | tail -1 | sed "s/^{/{\"lanes\": $L, /" >> $O/config5_lanes.jsonl
done
python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-group > $O/config5_nogroup.json 2>> $O/config5.err
python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --no-group --workers 16 > $O/config5_nogroup16.json 2>> $O/config5.err
timeout 600 tools/agroup_bench 32 > $O/agroup_bench_32.jsonl 2> $O/agroup_bench.err
timeout 600 tools/agroup_bench 8 > $O/agroup_bench_8.jsonl 2>> $O/agroup_bench.err
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
for l in open(o + "/config5_lanes.jsonl"):
    d = json.loads(l); print("lanes", d["lanes"], round(d["value"], 1), "comparisons/s  valu frac", round(d["roofline"]["frac"], 3), d.get("dispatcher"))
for f in ("config5_nogroup.json", "config5_nogroup16.json"):
    d = json.loads([l for l in open(o + "/" + f) if l.startswith("{")][0]); print(f, round(d["value"], 1))
PY
cat $O/agroup_bench_32.jsonl $O/agroup_bench_8.jsonl; tail -n 3 $O/*.err
