#!/bin/bash
# A/B of one environment switch on the C2 bench: bash tools/debug/ab_env.sh VAR [steps]   (runs VAR=0,1,0,1)
VAR=$1; STEPS=${2:-30}
for v in 0 1 0 1; do
  env $VAR=$v python bench.py --steps $STEPS --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'], d['value'])"
done
