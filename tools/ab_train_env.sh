#!/bin/bash
# Same-box A/B of the training step under two environments: tools/ab_train_env.sh "ENV_A" "ENV_B" [bench args] ("-" = no variables)
A="$1"; B="$2"; shift; shift
one() { local e="$1"; shift; [ "$e" = "-" ] && e=""; env $e python bench.py --mode train --steps 30 --warmup 5 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['ms_per_step'], 'ms', r['value'], 'img/s', 'loss', r.get('final_loss'))"; }
for i in 1 2; do echo "A[$A]: $(one "$A" "$@")"; echo "B[$B]: $(one "$B" "$@")"; done
