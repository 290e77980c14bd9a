#!/bin/bash
# tools/pmc_brick.sh — run ON THE GPU BOX: SQ instruction counters of the brick kernel in the chain (bench.py --lut-variant 6)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_brick
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra --lut-variant 6"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
  --output-format csv -d "$OUT/a" -o a -- $BENCH > "$OUT/a.log" 2>&1; echo "rc=$?"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- $BENCH > "$OUT/b.log" 2>&1; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/*counter_collection.csv") + glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        if "brick" not in k and "hsvfilter_flat_kernel<5, 0" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("   %-24s %.5g per launch" % (c, acc[k][c] / max(cnt[k][c], 1)))
PY
rm -rf "$OUT/a" "$OUT/b"
