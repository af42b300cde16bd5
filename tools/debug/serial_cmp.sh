for i in 1 2; do
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-stream', d['ms_per_step'])"
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs --serial 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('serial', d['ms_per_step'])"
done
