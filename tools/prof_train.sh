#!/bin/bash
# rocprofv3 kernel stats over the training bench. usage: bash tools/prof_train.sh <tag>
set -e
TAG=${1:-t}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG}_stats -- python3 bench.py --mode train --steps 4 --warmup 2 > gpurun_out/prof_${TAG}_stats.log 2>&1
find gpurun_out/prof_${TAG}_stats -name "*kernel_stats.csv" | head -3
