#!/bin/bash
# precision="fp16r32": accuracy and speed per set of two-pass convs. usage: bash tools/r32_sweep.sh "0 15 31 ..."
for m in $1; do
  echo "== mask $m"
  DMME_DEBUG_ROUTE=r32_2pass=$m python -m pytest tests/test_gpu_fp16.py -q -s -k "fp16r32 and not refusals" 2>&1 | grep -i "max|err|\|worst\|passed\|failed" | sed -e 's/; modules.*//' | cut -c1-200
  DMME_DEBUG_ROUTE=r32_2pass=$m python -m pytest tests/test_gpu_fp16.py -q -s -k "refusals" 2>&1 | grep -i "B=128" | cut -c1-200
  DMME_DEBUG_ROUTE=r32_2pass=$m python bench.py --no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 --steps 50 --warmup 10 --precision fp16r32 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps/s', d['value'], 'ms', d['ms_per_step'])"
done
