#!/bin/bash
# tools/pmc_colorlut.sh — run ON THE GPU BOX: SQ / LDS counters of one colorlut kernel variant (tools/run_colorlut_once.py).
#   bash tools/pmc_colorlut.sh <variant> [amp] [sets]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
V=${1:-7}; AMP=${2:-0}; SETS=${3:-32}
OUT=$R/gpurun_out/pmc_colorlut_v${V}_a${AMP}_s${SETS}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/a" -o a -- python3 $R/tools/run_colorlut_once.py $V 10 $AMP $SETS > "$OUT/a.log" 2>&1; echo "rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- python3 $R/tools/run_colorlut_once.py $V 10 $AMP $SETS > "$OUT/b.log" 2>&1; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True) + glob.glob(out + "/b/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:70]
        if "colorlut" not in k: continue
        name = row["Counter_Name"] + ("" if "/a/" in f or row["Counter_Name"] != "SQ_WAVE_CYCLES" else "#b")
        acc[k][name] += float(row["Counter_Value"])
        if name == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, d in acc.items():
    n = max(cnt[k], 1)
    print(k, "launches", n)
    for c in sorted(d): print("   %-24s %.4g per launch  (%.1f%% of WAVE_CYCLES)" % (c, d[c] / n, 100 * d[c] / max(d.get("SQ_WAVE_CYCLES", 1), 1)))
PY
