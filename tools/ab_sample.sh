#!/bin/bash
# Same-box A/B of the sampling step: tools/ab_sample.sh "ENV_A" "ENV_B" [batch] (each ENV is a space-separated list of VAR=value, or "-")
A="$1"; B="$2"; BATCH="${3:-128}"
Q="--no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 --steps 100 --warmup 20 --batch $BATCH"
one() { local e="$1"; [ "$e" = "-" ] && e=""; env $e python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for i in 1 2; do echo "A[$A]: $(one "$A")"; echo "B[$B]: $(one "$B")"; done
