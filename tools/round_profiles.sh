#!/bin/bash
# Everything the round's committed profiles / bench lines come from, in one GPU call.  usage: bash tools/round_profiles.sh <tag>
# Afterwards, locally: python tools/summarize_prof.py <tag> <label>; python tools/traffic.py <tag> <label>; copy the stats CSVs / JSON lines.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/prof.sh ${TAG} > gpurun_out/${TAG}_prof.log 2>&1
bash tools/prof_train.sh ${TAG}t > /dev/null 2>&1
bash tools/prof_sample.sh ${TAG}i --model iddpm64 --batch 32 > /dev/null 2>&1
bash tools/prof_train.sh ${TAG}it --model iddpm64 --batch 32 > /dev/null 2>&1
bash tools/prof_sample.sh ${TAG}x --precision bf16x3 > /dev/null 2>&1
bash tools/prof_sample.sh ${TAG}r --precision fp16r32 > /dev/null 2>&1
bash tools/prof_train.sh ${TAG}th --precision fp16 > /dev/null 2>&1
python bench.py --steps 100 --warmup 20 > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err
python bench.py --mode ddim --batch 512 --steps 50 --warmup 10 --no-cpu-baseline --no-accurate-leg --train-steps 0 > gpurun_out/${TAG}_bench_ddim_b512_n1.json 2>/dev/null
python bench.py --mode train --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_train_n1.json 2>/dev/null
python bench.py --model iddpm64 --batch 32 --steps 50 --warmup 10 --no-cpu-baseline --no-accurate-leg --train-steps 0 > gpurun_out/${TAG}_bench_iddpm64_b32_n1.json 2>/dev/null
python bench.py --model iddpm64 --batch 32 --mode train --steps 30 --warmup 5 > gpurun_out/${TAG}_bench_iddpm64_train_b32_n1.json 2>/dev/null
python bench.py --precision bf16x3 --steps 20 --warmup 5 --no-cpu-baseline --train-steps 0 > gpurun_out/${TAG}_bench_x3_n1.json 2>/dev/null
python bench.py --precision fp16r32 --steps 50 --warmup 10 --no-cpu-baseline --train-steps 0 > gpurun_out/${TAG}_bench_fp16r32_n1.json 2>/dev/null
python bench.py --model iddpm64 --batch 32 --precision fp16r32 --steps 50 --warmup 10 --no-cpu-baseline --no-accurate-leg --train-steps 0 > gpurun_out/${TAG}_bench_iddpm64_fp16r32_b32_n1.json 2>/dev/null
python bench.py --mode train --precision fp16 --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_train_fp16_n1.json 2>/dev/null
python bench.py --precision fp32 --steps 10 --warmup 3 --no-cpu-baseline --train-steps 0 > gpurun_out/${TAG}_bench_fp32_n1.json 2>/dev/null
for b in 1 32; do python bench.py --batch $b --steps 300 --warmup 30 --no-cpu-baseline --no-accurate-leg --train-steps 0 > gpurun_out/${TAG}_bench_b${b}_n1.json 2>/dev/null; done
ls -la gpurun_out/${TAG}_bench_*.json | awk '{print $5, $9}'
du -sh gpurun_out/prof_${TAG}* | tail -12
