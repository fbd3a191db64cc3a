#!/bin/bash
# rocprofv3 kernel stats over the training bench for several values of one env knob.
# usage: bash tools/prof_train_sweep.sh VAR v1 v2 ...
set -e
VAR=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
for v in "$@"; do
  export $VAR=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sweep_${VAR}_${v} -- python3 bench.py --mode train --steps 4 --warmup 2 > gpurun_out/sweep_${VAR}_${v}.log 2>&1
  f=$(find gpurun_out/sweep_${VAR}_${v} -name "*kernel_stats.csv" | head -1)
  echo "== $VAR=$v"; grep -E "wgrad" $f | cut -c1-60,100-260 | head -6
  tail -1 gpurun_out/sweep_${VAR}_${v}.log | cut -c100-135
done
