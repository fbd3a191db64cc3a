#!/bin/bash
# After `gpurun -- bash tools/round_profiles.sh <tag>`: condense gpurun_out/prof_<tag>* into the tracked files under profiles/.
# usage: bash tools/collect_round.sh r04
set -e
TAG=${1:-r04}
cd "$(dirname "$0")/.."
python tools/summarize_prof.py ${TAG} ${TAG}_sample_b128_bf16 > /dev/null
python tools/traffic.py ${TAG} ${TAG}_sample_b128_bf16 > /dev/null
python tools/mfma_busy.py ${TAG} ${TAG}_sample_b128_bf16 > /dev/null
newest() { ls -t $1 2>/dev/null | head -1; }
cp "$(newest "gpurun_out/prof_${TAG}t_stats/*/*kernel_stats.csv")" profiles/${TAG}_train_b128_bf16_kernel_stats.csv
cp "$(newest "gpurun_out/prof_${TAG}i_stats/*/*kernel_stats.csv")" profiles/${TAG}_sample_iddpm64_b32_bf16_kernel_stats.csv
cp "$(newest "gpurun_out/prof_${TAG}it_stats/*/*kernel_stats.csv")" profiles/${TAG}_train_iddpm64_b32_bf16_kernel_stats.csv
cp "$(newest "gpurun_out/prof_${TAG}x_stats/*/*kernel_stats.csv")" profiles/${TAG}_sample_b128_bf16x3_kernel_stats.csv
cp "$(newest "gpurun_out/prof_${TAG}r_stats/*/*kernel_stats.csv")" profiles/${TAG}_sample_b128_fp16r32_kernel_stats.csv
cp "$(newest "gpurun_out/prof_${TAG}th_stats/*/*kernel_stats.csv")" profiles/${TAG}_train_b128_fp16_kernel_stats.csv
for f in gpurun_out/${TAG}_bench_*.json; do [ -s "$f" ] && cp "$f" profiles/; done
ls -la profiles/${TAG}_* | awk '{print $5, $9}'
