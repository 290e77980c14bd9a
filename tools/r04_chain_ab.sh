#!/bin/bash
# round 4, GPU call: the chain's colorlut kernel in its real place (input left in the Infinity Cache by hsvfilter): table
# kernels A/B through bench.py on one box
out=gpurun_out/r04j; mkdir -p $out
for v in 5 9 8 5 8; do
  python bench.py --lut-variant $v --no-extra --no-cpu-baseline --no-live-pmc --steps 40 > $out/b_$v.json 2>/dev/null
  python - $v $out/b_$v.json <<'PY' | tee -a $out/chain_ab.txt
import json,sys
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels"]
print("variant",sys.argv[1],"value %.0f"%d["value"],"hsv %.4f lut %.4f"%(k["hsvfilter_ms_per_launch"],k["colorlut_ms_per_launch"]),k["colorlut_kernels_served"])
PY
done
