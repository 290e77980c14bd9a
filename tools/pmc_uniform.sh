#!/bin/bash
# tools/pmc_uniform.sh — run ON THE GPU BOX: SQ / LDS counters of the whole-plane interpolating kernel (colorlut3d_lds_kernel, LUT variant 3)
# on 8 x 4K UNIFORM NOISE and, for contrast, on natural-like frames: is the LDS pipe what uniform noise saturates?
# (profiles/r05_uniform_bound.txt = its output + the cycle arithmetic.)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_uniform
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY=$(python3 -c 'import os, sys; print(os.path.realpath(sys.executable))')
for amp in -1 0; do
  RUN="$PY $R/tools/run_colorlut_once.py 3 8 $amp"
  tag=amp$amp
  $RUN | tail -1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES \
    --output-format csv -d "$OUT/${tag}a" -o a -- $RUN > "$OUT/${tag}a.log" 2>&1; echo "$tag a rc=$?"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/${tag}b" -o b -- $RUN > "$OUT/${tag}b.log" 2>&1; echo "$tag b rc=$?"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
for tag in sorted(set(os.path.basename(d)[:-1] for d in glob.glob(out + "/amp*[ab]") if os.path.isdir(d))):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
    for f in glob.glob(out + "/" + tag + "?/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0][:70]
            if "colorlut3d_lds_kernel" not in k: continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[k][row["Counter_Name"]] += 1
    for k in sorted(acc):
        print("%s (%s)  %s" % (tag, "uniform noise" if tag == "amp-1" else "natural-like", k))
        for c in sorted(acc[k]): print("   %-24s %.5g per launch" % (c, acc[k][c] / max(cnt[k][c], 1)))
        a = {c: acc[k][c] / max(cnt[k][c], 1) for c in acc[k]}
        wp = 8 * 3840 * 2160 / 64.0
        if "SQ_INSTS_VALU" in a: print("   per wave-pixel (64 pixels): VALU %.1f  LDS %.2f  SALU %.1f instructions" % (a["SQ_INSTS_VALU"] / wp, a.get("SQ_INSTS_LDS", 0) / wp, a.get("SQ_INSTS_SALU", 0) / wp))
        if "SQ_LDS_IDX_ACTIVE" in a: print("   LDS index cycles per LDS instruction %.2f, of which bank-conflict cycles %.2f ; LDS-wait share of wave cycles %.3f" % (a["SQ_LDS_IDX_ACTIVE"] / max(a.get("SQ_INSTS_LDS", 1), 1), a.get("SQ_LDS_BANK_CONFLICT", 0) / max(a.get("SQ_INSTS_LDS", 1), 1), a.get("SQ_WAIT_INST_LDS", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)))
PY
rm -rf "$OUT"/amp*a "$OUT"/amp*b
