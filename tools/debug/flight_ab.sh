run() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
run "depth 2 (default)"
SUHPE_STEPS_IN_FLIGHT=1 run "depth 1"
SUHPE_STEPS_IN_FLIGHT=3 run "depth 3"
done
