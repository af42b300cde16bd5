# A/B on one box: the committed tile database against another (SUHPE_TILE_DB), alternating
B=${1:-tools/debug/tiles_prev.json}
for i in 1 2 3; do
python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('A (tree) ', d['ms_per_step'])"
SUHPE_TILE_DB=$B python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B (other)', d['ms_per_step'])"
done
