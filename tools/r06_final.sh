#!/bin/bash
# tools/r06_final.sh — run ON THE GPU BOX on the final code of the round: the GPU suite, the driver's protocol, the all-legs kernel trace
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_final; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/r06_final_pytest.txt 2>&1; echo "pytest rc=$?" >> $O/r06_final_pytest.txt
tail -4 $O/r06_final_pytest.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r06_driver_protocol.json 2> $O/driver_protocol.err
bash tools/trace_full_once.sh > $O/trace_full_once.log 2>&1
cp gpurun_out/trace_full_once/kernel_stats_full.csv $O/r06_kernel_stats_full.csv 2>/dev/null
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/r06_driver_protocol.json") if l.startswith("{")][0])
print("driver protocol:", round(d["value"]), "frac", round(d["roofline"]["frac"], 4), d["roofline"].get("kernel"), "traffic", d["roofline"].get("traffic"), "fused", round(d["fused_chain"]["frames_per_s"]))
c5 = d["config5"]; print("  config5", round(c5["comparisons_per_s"]), "dispatcher", round(c5["through_the_dispatcher"]["comparisons_per_s"]), "native", c5.get("native_element_threads", {}).get("dispatcher_comparisons_per_s"))
PY
head -6 $O/r06_kernel_stats_full.csv | cut -c1-160
