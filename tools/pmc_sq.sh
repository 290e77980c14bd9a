#!/bin/bash
# tools/pmc_sq.sh — run ON THE GPU BOX: one rocprofv3 --pmc pass with SQ counters over the default bench workload
# (wave-time split of the kernels: parked vs issue-stalled vs active, LDS activity/conflicts).
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extra"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/a" -o a -- $BENCH > "$OUT/a.log" 2>&1; echo "rc=$?"
# second pass: texture-address / L1 activity (what the table kernel waits on)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- $BENCH > "$OUT/b.log" 2>&1; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/b/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        name = row["Counter_Name"] + ("" if "/a/" in f or row["Counter_Name"] not in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY") else "#b")
        acc[k][name] += float(row["Counter_Value"])
        if name == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in acc.items():
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c in sorted(d): print("   %-24s %.4g per launch  (%.1f%% of WAVE_CYCLES)" % (c, d[c] / n, 100 * d[c] / max(d.get("SQ_WAVE_CYCLES", 1), 1)))
PY
