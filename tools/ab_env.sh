#!/bin/bash
# same-box A/B of (library, environment) pairs: tools/ab_env.sh "lib1.so|ENV=.. lib2.so|- ..." [rounds] [sample|train|both] [bench args]
SPECS="$1"; R=${2:-2}; WHAT=${3:-sample}; shift; shift; shift
Q="--no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 --steps 100 --warmup 20"
for r in $(seq $R); do
  for sp in $SPECS; do
    lib=${sp%%|*}; e=${sp#*|}; [ "$e" = "-" ] && e=""; e=${e//,/ }
    if [ "$WHAT" != "train" ]; then
      env $e DMME_LIB_PATH=$PWD/$lib python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sample', '$sp', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
    fi
    if [ "$WHAT" != "sample" ]; then
      env $e DMME_LIB_PATH=$PWD/$lib python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('train ', '$sp', r['ms_per_step'], 'ms', r['value'], 'img/s', 'loss', r['final_loss'])"
    fi
  done
done
