#!/bin/bash
# Same-box A/B of several builds of the library (DMME_LIB_PATH): tools/ab_libs.sh "lib1.so lib2.so ..." [rounds] [sample|train|both] [extra bench args]
LIBS="$1"; R=${2:-2}; WHAT=${3:-sample}; shift; shift; shift
Q="--no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 --steps 100 --warmup 20"
for r in $(seq $R); do
  for lib in $LIBS; do
    if [ "$WHAT" != "train" ]; then
      DMME_LIB_PATH=$PWD/$lib python bench.py $Q "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sample', '$lib', d['value'], 'steps/s', d['ms_per_step'], 'ms')"
    fi
    if [ "$WHAT" != "sample" ]; then
      DMME_LIB_PATH=$PWD/$lib python bench.py --mode train --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('train ', '$lib', r['ms_per_step'], 'ms', r['value'], 'img/s', 'loss', r['final_loss'])"
    fi
  done
done
