#!/bin/bash
# GPU call: the whole GPU suite, then the default bench line (summary to gpurun_out/$1)
out=gpurun_out/${1:-r04_full}; mkdir -p $out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $out/gpu_tests.txt
python bench.py > $out/bench.json 2> $out/bench.err
python tools/bench_summary.py $out/bench.json | tee $out/bench_summary.txt
