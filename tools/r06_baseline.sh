#!/bin/bash
# tools/r06_baseline.sh — run ON THE GPU BOX (round 6, first call): the state the round starts from, on one box:
# driver's protocol, the fused leg at 16 frames per launch, fresh SQ passes for the brick kernel and the Dssim kernels,
# HBM traffic counters for the Dssim kernels (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), walk_bench.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_baseline; rm -rf $O; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_protocol.json 2> $O/driver_protocol.err
python3 bench.py --steps 20 --warmup 5 --batch 16 --ring 2 --no-cpu-baseline --no-live-pmc > $O/batch16.json 2> $O/batch16.err
bash tools/pmc_brick.sh > $O/brick_sq_counters.txt 2>&1
bash tools/pmc_dssim.sh > $O/dssim_sq_counters.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$c -o p -- python3 $R/tools/dssim_once.py 3 > $R/$O/pmc_$c.log 2>&1
  done )
python3 - $O <<'PY' > $O/dssim_traffic.txt 2>&1
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:48] + " grid=" + row["Grid_Size"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k, {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]})
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python3 tools/bench_dssim.py > $O/bench_dssim.json 2>&1
python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline > $O/config5.json 2> $O/config5.err
python3 bench.py --config 5 --steps 20 --warmup 5 --no-cpu-baseline --shared-reference --workers 2 > $O/config5_shared2.json 2> $O/config5_shared2.err
ls -la $O
