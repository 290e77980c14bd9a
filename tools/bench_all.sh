#!/bin/bash
# tools/bench_all.sh <tag> — run ON THE GPU BOX (via gpurun): every per-config bench tool, outputs collected into
# gpurun_out/configs_<tag>.jsonl (one JSON object per line, tagged with the tool name). Copy into profiles/ afterwards:
#   cp gpurun_out/configs_<tag>.jsonl gpurun_out/configs_<tag>_elements.txt profiles/
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/configs_$TAG.jsonl
mkdir -p "$R/gpurun_out"; : > "$OUT"
python3 "$R/tools/bench_elements.py" 2>/dev/null > "$R/gpurun_out/configs_${TAG}_elements.txt"
for t in bench_hrtf bench_videocompare bench_dssim bench_loudnorm bench_loudnorm_streams bench_ebur128 bench_echo bench_pipeline bench_streams; do
  echo "== $t" >&2
  python3 "$R/tools/$t.py" 2>/dev/null | grep '^{' | tail -1 | sed "s/^{/{\"tool\": \"$t\", /" >> "$OUT"
done
python3 "$R/tools/bench_hrtf.py" --taps 512 --no-cpu 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench_hrtf_512taps", /' >> "$OUT"
python3 "$R/bench.py" --config 5 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5", /' >> "$OUT"
python3 "$R/bench.py" --config 5 --group 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5 --group", /' >> "$OUT"
python3 "$R/bench.py" --config 5 --hash-algo blockhash 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5 --hash-algo blockhash", /' >> "$OUT"
python3 "$R/bench.py" --config 5 --hash-algo blockhash --group 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5 --hash-algo blockhash --group", /' >> "$OUT"
for n in 32 8 2; do "$R/tools/agroup_bench" $n 2>/dev/null | grep '^{' | sed "s/^{/{\"tool\": \"agroup_bench $n (native threads)\", /" >> "$OUT"; done
python3 "$R/bench.py" --config 5 --shared-reference --workers 2 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5 --shared-reference --workers 2", /' >> "$OUT"
python3 "$R/bench.py" --config 5 --shared-reference --workers 4 --dssim-two-step 2>/dev/null | grep '^{' | tail -1 | sed 's/^{/{"tool": "bench.py --config 5 --shared-reference --workers 4 --dssim-two-step", /' >> "$OUT"
python3 "$R/tools/bench_rgba64.py" 2>/dev/null >> "$R/gpurun_out/configs_${TAG}_elements.txt"
python3 "$R/tools/bench_loudnorm_batch.py" 2>/dev/null >> "$R/gpurun_out/configs_${TAG}_elements.txt"
python3 "$R/tools/bench_sofa.py" 2>/dev/null | grep '^{' | sed 's/^{/{"tool": "bench_sofa", /' >> "$OUT"
wc -l "$OUT" >&2
