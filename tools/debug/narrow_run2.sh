python -m pytest tests/test_gpu_winograd.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-stream', d['ms_per_step'], d['value'])"; done
