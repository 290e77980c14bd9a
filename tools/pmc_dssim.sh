#!/bin/bash
# tools/pmc_dssim.sh — run ON THE GPU BOX: SQ counters of the Dssim kernels (instruction mix and wave-time split), 4K frames
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_dssim
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
  --output-format csv -d "$OUT/a" -o a -- python3 $R/tools/dssim_once.py 3 > "$OUT/a.log" 2>&1; echo "rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d "$OUT/b" -o b -- python3 $R/tools/dssim_once.py 3 > "$OUT/b.log" 2>&1; echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
# per kernel name: keep only the LARGEST launch class (scale 0) by grid size
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:48] + " grid=" + row["Grid_Size"]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    d = acc[k]
    print(k)
    for c in sorted(d):
        n = max(cnt[k][c], 1)
        print("   %-24s %.5g per launch" % (c, d[c] / n))
PY
