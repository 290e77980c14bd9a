#!/bin/bash
# run ON THE GPU BOX: everything profiles/r03_* is made from, in one call (final state of the round). Only gpurun_out/ travels
# back (<= 64 MiB): the summaries are made here, copied to gpurun_out/final_profiles/, and the raw traces are deleted.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -rf gpurun_out/profiles_r03 gpurun_out/profiles_r03_interp gpurun_out/prof_dssim gpurun_out/pmc_dssim gpurun_out/final_profiles
mkdir -p gpurun_out/final_profiles
bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1
python3 tools/summarize_profiles.py r03 > gpurun_out/summarize_r03.log 2>&1
cp profiles/pmc_latest.json gpurun_out/final_profiles/pmc_latest.json
bash tools/collect_profiles.sh r03_interp "--lut-variant 6" > gpurun_out/collect_r03_interp.log 2>&1
cp profiles/pmc_latest.json /tmp/pmc_keep.json
python3 tools/summarize_profiles.py r03_interp > gpurun_out/summarize_r03_interp.log 2>&1
cp /tmp/pmc_keep.json profiles/pmc_latest.json     # pmc_latest stays the default command's
bash tools/prof_dssim.sh > gpurun_out/final_profiles/r03_dssim_kernel_stats.txt 2>&1
bash tools/pmc_dssim.sh > gpurun_out/final_profiles/r03_dssim_sq_counters.txt 2>&1
cp profiles/r03_* gpurun_out/final_profiles/
rm -rf gpurun_out/profiles_r03 gpurun_out/profiles_r03_interp gpurun_out/prof_dssim gpurun_out/pmc_dssim
bash tools/bench_all.sh r03 > gpurun_out/bench_all_r03.log 2>&1
cp gpurun_out/configs_r03.jsonl gpurun_out/final_profiles/r03_configs.jsonl
cp gpurun_out/configs_r03_elements.txt gpurun_out/final_profiles/r03_configs_elements.txt
du -sh gpurun_out; ls gpurun_out/final_profiles; tail -2 gpurun_out/summarize_r03.log
