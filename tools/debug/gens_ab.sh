run() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['other_configs'].items()})"; }
for i in 1 2; do
run "gens 1"
SUHPE_WINO_GENS=2 run "gens 2"
SUHPE_WINO_GENS=4 run "gens 4"
done
