#!/bin/bash
# tools/final_collect.sh <tag> — run ON THE GPU BOX: everything profiles/<tag>_* is made from, in one call (final state of a
# round). Only gpurun_out/ travels back (<= 64 MiB): the summaries are made here, copied to gpurun_out/final_profiles/, and the
# raw traces are deleted.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
F=gpurun_out/final_profiles
rm -rf gpurun_out/profiles_$TAG gpurun_out/profiles_${TAG}_interp gpurun_out/profiles_${TAG}_window gpurun_out/prof_dssim gpurun_out/pmc_dssim $F
mkdir -p $F
bash tools/collect_profiles.sh $TAG > gpurun_out/collect_$TAG.log 2>&1
python3 tools/summarize_profiles.py $TAG > gpurun_out/summarize_$TAG.log 2>&1
cp profiles/pmc_latest.json $F/pmc_latest.json
python3 tools/launch_gaps.py gpurun_out/profiles_$TAG/trace > $F/${TAG}_launch_gaps.txt 2>&1
cp profiles/pmc_latest.json /tmp/pmc_keep.json
bash tools/collect_profiles.sh ${TAG}_interp "--lut-variant 6" > gpurun_out/collect_${TAG}_interp.log 2>&1
python3 tools/summarize_profiles.py ${TAG}_interp > gpurun_out/summarize_${TAG}_interp.log 2>&1
# the LDS-cached table kernel pinned as the chain's colorlut (variant 8): what auto turns down behind hsvfilter
bash tools/collect_profiles.sh ${TAG}_window "--lut-variant 8" > gpurun_out/collect_${TAG}_window.log 2>&1
python3 tools/summarize_profiles.py ${TAG}_window > gpurun_out/summarize_${TAG}_window.log 2>&1
cp /tmp/pmc_keep.json profiles/pmc_latest.json     # pmc_latest stays the default command's
python3 tools/window_probe.py 0 4 8 16 > $F/${TAG}_window_probe.txt 2>&1
bash tools/prof_dssim.sh > $F/${TAG}_dssim_kernel_stats.txt 2>&1
bash tools/pmc_dssim.sh > $F/${TAG}_dssim_sq_counters.txt 2>&1
bash tools/pmc_brick.sh > $F/${TAG}_brick_sq_counters.txt 2>&1
timeout 300 tools/walk_dma_bench > $F/${TAG}_walk_dma_bench_box.txt 2>&1
timeout 300 python3 tools/stress_dispatch.py 25 > $F/${TAG}_stress_dispatch.txt 2>&1
cp -n profiles/${TAG}_* $F/     # (what the steps above wrote into $F stays: -n)
rm -rf gpurun_out/profiles_$TAG gpurun_out/profiles_${TAG}_interp gpurun_out/profiles_${TAG}_window gpurun_out/prof_dssim gpurun_out/pmc_dssim gpurun_out/pmc_brick
bash tools/bench_all.sh $TAG > gpurun_out/bench_all_$TAG.log 2>&1
cp gpurun_out/configs_$TAG.jsonl $F/${TAG}_configs.jsonl
cp gpurun_out/configs_${TAG}_elements.txt $F/${TAG}_configs_elements.txt
du -sh gpurun_out; ls $F; tail -2 gpurun_out/summarize_$TAG.log
