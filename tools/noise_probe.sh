#!/bin/bash
# tools/noise_probe.sh — run ON THE GPU BOX: the interpolating colorlut kernels against noise amplitude (bench.py's batches: one
# natural-like base frame + uniform noise of +-amp per channel, eight rotations per batch), alone and in the chain, per launch
# size; SQ counters of the shared-cache and per-wave kernels; per-block timing of the shared-cache kernel (debug build, last).
# Writes gpurun_out/noise_probe.txt (copied to profiles/rNN_noise_probe.txt by hand, with the round's notes on top).
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}" || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/noise_probe.txt
{
echo "== colorlut alone (same pre-filtered batch every launch, device kept busy), 8 x 4K per launch: ms per launch (median of 80; min), share of"
echo "   256-pixel steps with a cache miss / served from memory. v7/32, v7/64 = per-wave caches pinned, v7/512 = block-shared cache pinned,"
echo "   v3 = three-pass whole-plane kernel, v6 = content watch"
MODE=lutsame AMPS=0,2,4,8,12,16,20,24,32 CONFIGS=7:32,7:64,7:512,3:0,6:0 timeout 1200 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids
echo
echo "== in the chain (copy of a pristine batch, hsvfilter in place, colorlut), 8 x 4K per launch"
MODE=chain AMPS=0,4,8,16 CONFIGS=7:32,7:64,7:512,3:0,6:0 timeout 1200 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids
for n in 1 2; do
echo
echo "== colorlut alone, $n x 4K per launch"
N=$n MODE=lutsame AMPS=0,4,8,16 CONFIGS=7:32,7:64,7:512,3:0,6:0 timeout 1200 python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids
done
echo
echo "== SQ counters (rocprofv3 --pmc, three passes; tools/run_colorlut_once.py: 8 x 4K, eight different frames), per launch"
CASES="512:0 512:8 32:0 64:8" bash tools/pmc_shared.sh 2>&1 | grep -v amdgpu.ids
echo
echo "== per-block timing of colorlut3d_shared_kernel (-DBRICK_TIMING build; rotated frames as in bench.py), amp 0 and 8"
bash tools/exp_brick_build.sh "-DBRICK_TIMING"
for amp in 0 8; do
  ROT=1 BRICK_TIMING_FILE=/tmp/t.bin python tools/run_colorlut_once.py 7 3 $amp 512 2>&1 | grep -v amdgpu.ids
  python tools/shared_timing.py /tmp/t.bin
done
} > $OUT 2>&1
tail -5 $OUT
