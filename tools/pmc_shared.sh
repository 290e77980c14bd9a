#!/bin/bash
# tools/pmc_shared.sh — run ON THE GPU BOX: SQ counters of the interpolating kernels alone (tools/run_colorlut_once.py), per cache
# geometry and noise amplitude: "sets:amp" pairs in $CASES (default: shared cache and the per-wave caches at amp 0 and 8)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_shared
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in ${CASES:-512:0 512:8 32:0 64:8}; do
  sets=${c%%:*}; amp=${c##*:}
  RUN="python3 $R/tools/run_colorlut_once.py 7 10 $amp $sets"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    --output-format csv -d "$OUT/a_$sets_$amp" -o a -- $RUN > "$OUT/a.log" 2>&1
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/b_$sets_$amp" -o b -- $RUN > "$OUT/b.log" 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INSTS_BRANCH SQ_INSTS_SENDMSG \
    --output-format csv -d "$OUT/c_$sets_$amp" -o c -- $RUN > "$OUT/c.log" 2>&1
  echo "== sets $sets amp $amp"
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:70]
        if "brick" not in k and "shared" not in k: continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("   %-24s %.5g per launch" % (c, acc[k][c] / max(cnt[k][c], 1)))
PY
  rm -rf "$OUT"/a_* "$OUT"/b_* "$OUT"/c_*
done
