#!/bin/bash
# round 4, GPU call: the window kernel's floors in the chain (input from the Infinity Cache): copy walk, hit path, full
out=gpurun_out/r04l; mkdir -p $out
run() { python bench.py --lut-variant $1 --no-extra --no-cpu-baseline --no-live-pmc --steps 40 > $out/b.json 2>/dev/null
  python - "$2" $out/b.json <<'PY' | tee -a $out/chain_ab2.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[1],"value %.0f"%d["value"],"hsv %.4f lut %.4f"%(k["hsvfilter_ms_per_launch"],k["colorlut_ms_per_launch"]),k["colorlut_kernels_served"])
PY
}
run 5 "gather kernel (variant 5)"
for flags in "" "-DWIN_EXP=2" "-DWIN_EXP=1" "-DWIN_FILLS=16" "-DWIN_DEPTH=1" "-DWIN_DEPTH=3"; do
  tools/exp_window_build.sh "$flags"
  run 8 "window kernel, flags: $flags"
done
tools/exp_window_build.sh ""
run 5 "gather kernel (variant 5)"
