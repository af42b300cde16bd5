run() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['config'].get('loss_all'), d['config']['hbm_gib'])"; }
for i in 1 2 3; do
run "default (main high)"
SUHPE_MAIN_PRIO=0 run "callers stream     "
done
