run() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
run "both        "
SUHPE_OVERLAP=0 run "wgrad only  "
SUHPE_WGRAD_OVERLAP=0 run "teacher only"
SUHPE_OVERLAP=0 SUHPE_WGRAD_OVERLAP=0 run "none        "
done
