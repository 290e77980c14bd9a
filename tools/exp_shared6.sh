#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out
for which in brick chain; do
  rm -rf /tmp/kt_$which
  if [ $which = brick ]; then
    VARIANTS=7 SETS=512 AMPS=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$which -o kt -- python3 $R/tools/bench_brick.py > /tmp/kt_$which.log 2>&1
  else
    MODE=lutsame AMPS=0 CONFIGS=7:512 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_$which -o kt -- python3 $R/tools/chain_probe.py > /tmp/kt_$which.log 2>&1
  fi
  tail -2 /tmp/kt_$which.log
  python3 - /tmp/kt_$which <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sh = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "shared" in r["Kernel_Name"]]
d = sorted(e - s for s, e in sh)
gaps = sorted(sh[i + 1][0] - sh[i][1] for i in range(len(sh) - 1))
print("shared kernel launches %d: duration us min %.1f median %.1f p90 %.1f max %.1f; gap to next median %.1f us" % (len(d), d[0] / 1e3, d[len(d) // 2] / 1e3, d[int(len(d) * .9)] / 1e3, d[-1] / 1e3, gaps[len(gaps) // 2] / 1e3))
print("last 12 durations:", [round((e - s) / 1e3, 1) for s, e in sh[-12:]])
PY
done 2>&1 | tee $R/gpurun_out/shared_kt.log
