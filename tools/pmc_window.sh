#!/bin/bash
# tools/pmc_window.sh — run ON THE GPU BOX: SQ instruction / LDS counters of the memoised-table kernels on 8 x 4K natural-like frames
# from HBM: colorlut_window_kernel (variant 8; ORDER=0 contiguous shares, ORDER=1 aligned fronts) next to the gather kernel (variant 5).
# Two passes per kernel (the counters do not fit one). Prints per-launch averages; profiles/r05_window_sq_counters.txt is its output.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_window
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c 'import sys; print(sys.executable)')
for cfg in "8 0" "8 1" "5 0"; do
  set -- $cfg
  export ORDER=$2
  RUN="$PY $R/tools/run_colorlut_once.py $1 12 0"
  tag=v$1o$2
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    --output-format csv -d "$OUT/${tag}a" -o a -- $RUN > "$OUT/${tag}a.log" 2>&1; echo "$tag a rc=$?"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/${tag}b" -o b -- $RUN > "$OUT/${tag}b.log" 2>&1; echo "$tag b rc=$?"
  grep -h "ms per launch" "$OUT/${tag}a.log" "$OUT/${tag}b.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in sorted(set(os.path.basename(d)[:-1] for d in glob.glob(out + "/v*[ab]") if os.path.isdir(d))):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(out + "/" + tag + "?/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:70]
            if "colorlut_window_kernel" not in k and "colorlut_table_tiled_kernel" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
    for k in sorted(acc):
        print("%s  %s" % (tag, k))
        for c in sorted(acc[k]): print("   %-24s %.5g per launch" % (c, acc[k][c] / max(cnt[k][c], 1)))
        a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
        wp = 8 * 3840 * 2160 / 64.0
        if "SQ_INSTS_VALU" in a: print("   per wave-pixel (64 pixels): VALU %.1f  LDS %.2f  SALU %.1f instructions" % (a["SQ_INSTS_VALU"] / wp, a.get("SQ_INSTS_LDS", 0) / wp, a.get("SQ_INSTS_SALU", 0) / wp))
        if "SQ_WAIT_INST_LDS" in a and "SQ_WAVE_CYCLES" in a: print("   LDS-wait share of wave cycles %.3f ; LDS busy (SQ_ACTIVE_INST_LDS / SQ_BUSY_CYCLES) %.3f ; bank-conflict cycles / LDS index cycles %.3f" % (a["SQ_WAIT_INST_LDS"] / a["SQ_WAVE_CYCLES"], a.get("SQ_ACTIVE_INST_LDS", 0) / max(a.get("SQ_BUSY_CYCLES", 1), 1), a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_LDS_IDX_ACTIVE", 1), 1)))
PY
rm -rf "$OUT"/v*a "$OUT"/v*b
