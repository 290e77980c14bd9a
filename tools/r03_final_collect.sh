#!/bin/bash
# run ON THE GPU BOX: everything profiles/r03_* is made from, in one call (final state of the round)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1
bash tools/collect_profiles.sh r03_interp "--lut-variant 6" > gpurun_out/collect_r03_interp.log 2>&1
bash tools/prof_dssim.sh > gpurun_out/prof_dssim_summary.txt 2>&1
bash tools/pmc_dssim.sh > gpurun_out/pmc_dssim_summary.txt 2>&1
bash tools/bench_all.sh r03 > gpurun_out/bench_all_r03.log 2>&1
tail -3 gpurun_out/collect_r03.log; tail -2 gpurun_out/collect_r03_interp.log; head -12 gpurun_out/prof_dssim_summary.txt
