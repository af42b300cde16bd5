run() { python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'])"; }
for i in 1 2; do
run "tree                    "
SUHPE_TILE_DB=tools/debug/db_v1.json run "wgrad direct            "
SUHPE_TILE_DB=tools/debug/db_v2.json run "teacher fwd direct      "
SUHPE_TILE_DB=tools/debug/db_v3.json run "dgrad direct            "
done
