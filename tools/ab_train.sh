#!/bin/bash
# same-box A/B of two builds of the library on the training step: DMME_LIB_PATH selects the build.
# usage (on the GPU box): bash tools/ab_train.sh <prev.so> [rounds] [bench args]
PREV=$1; R=${2:-2}; shift; shift
for r in $(seq $R); do
  for lib in "$PREV" ""; do
    tag=$([ -z "$lib" ] && echo new || echo prev)
    DMME_LIB_PATH=$lib python bench.py --mode train --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('train', '$tag', r['ms_per_step'], 'ms', r['value'], 'img/s', 'loss', r['final_loss'])"
  done
done
