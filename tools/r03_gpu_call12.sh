#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r03l; mkdir -p $O
timeout 300 python bench.py --config 5 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pairs, 8 workers:', d['value'])"
for w in 1 2 4; do timeout 300 python bench.py --config 5 --steps 30 --warmup 3 --no-cpu-baseline --shared-reference --workers $w 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('shared reference, workers $w:', d['value'])"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o d -- python3 $GRAFT_REPO_ROOT/tools/bench_dssim.py > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
