#!/bin/bash
# round 4, GPU call: window kernel on Morton tables + nested auto choice: parity, then the bench line
out=gpurun_out/r04k; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_window.py tests/test_gpu_parity.py -x -q -k "window or table or auto or fused" 2>&1 | tail -8 | tee $out/tests.txt
python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python - $out/bench.json <<'PY' | tee $out/bench_summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",d["value"],"ms/step",d["ms_per_step"])
print("roofline",{k:d["roofline"][k] for k in ("kernel","frac","avg_launch_ms","traffic","rocprof_avg_launch_ms","traffic_live_note")})
print("kernels",d["kernels"]["hsvfilter_ms_per_launch"],d["kernels"]["colorlut_ms_per_launch"],d["kernels"]["colorlut_kernels_served"])
print("interp",d["interpolating_kernel_only"]["frames_per_s"])
print("fused",d["fused_chain"]["frames_per_s"],d["fused_chain"]["ms_per_launch"],d["fused_chain"]["kernel"],d["fused_chain"]["kernels_served"])
print("noise",d["other_content"]["frames_per_s"])
print("streams",d["concurrent_streams"]["frames_per_s"],d["concurrent_streams"]["colorlut_kernel"])
for k,v in d["content_sweep"].items():
    print(k,v["auto"]["frames_per_s"],v["auto"]["colorlut_ms_per_launch"],v["auto"]["colorlut_kernels_served"],"| interp",v["interpolating"]["frames_per_s"])
print("config5",d["config5"]["comparisons_per_s"])
PY
