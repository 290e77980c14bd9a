#!/bin/bash
# tools/r06_check3.sh — run ON THE GPU BOX: the GPU suite, both dispatcher stress tools, the audio groups from native threads, the driver's protocol
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
O=gpurun_out/r06_check3; rm -rf $O; mkdir -p $O
timeout 1500 python3 -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
tail -8 $O/pytest_gpu.txt
timeout 300 python3 tools/stress_dispatch.py 25 > $O/stress_dispatch.txt 2>&1; echo "stress rc=$?" >> $O/stress_dispatch.txt; tail -5 $O/stress_dispatch.txt
timeout 300 ./tools/agroup_bench 32 audio > $O/agroup_bench_32.jsonl 2> $O/agroup_bench.err; cut -c1-420 $O/agroup_bench_32.jsonl
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_protocol.json 2> $O/driver_protocol.err
python3 - $O <<'PY'
import json, sys
o = sys.argv[1]
d = json.loads([l for l in open(o + "/driver_protocol.json") if l.startswith("{")][0])
print("driver protocol:", round(d["value"]), "frac", round(d["roofline"]["frac"], 4), d["roofline"].get("kernel"), "fused", round(d["fused_chain"]["frames_per_s"]), d["fused_chain"].get("kernels_served"))
for k, v in d["content_sweep"].items():
    print("  sweep", k, {a: (round(b["frames_per_s"]), b["colorlut_kernels_served"]) for a, b in v.items() if isinstance(b, dict) and "frames_per_s" in b})
c5 = d["config5"]; print("  config5", round(c5["comparisons_per_s"]), "dispatcher", round(c5["through_the_dispatcher"]["comparisons_per_s"]), "valu frac", round(c5["roofline"]["frac"], 3), "native", c5.get("native_element_threads"))
print("  streams", json.dumps(d["concurrent_streams"])[:900])
PY
tail -n 3 $O/*.err
