#!/bin/bash
# rocprofv3 kernel stats over the sampling bench. usage: bash tools/prof_sample.sh <tag> [extra bench.py args]
set -e
TAG=${1:-s}
shift || true
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 "$@" > gpurun_out/prof_${TAG}_stats.log 2>&1
find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -3
