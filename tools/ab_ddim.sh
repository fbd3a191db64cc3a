one() { env $1 python bench.py --mode ddim --batch 512 --steps 50 --warmup 10 --no-cpu-baseline --no-accurate-leg --no-roofline --train-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
for e in X=1 DMME_DEBUG_ROUTE=lvl_max_iter=4 X=1 DMME_DEBUG_ROUTE=lvl_max_iter=4; do echo "$e: $(one $e)"; done
