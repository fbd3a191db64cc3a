Q="--no-cpu-baseline --no-roofline --no-accurate-leg --no-ddim-leg --no-small-batch-leg --reps 1 --train-steps 0 --steps 100 --warmup 20 --batch 128"
one() { env $1 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for e in X=1 DMME_NO_ATTN_FULL=1 X=1 DMME_NO_ATTN_FULL=1; do echo "$e: $(one $e)"; done
