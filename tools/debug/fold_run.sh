python -m pytest tests/test_gpu_winograd.py -m gpu -x -q 2>&1 | tail -5
WINO_SHAPES="48,192,1,56,160,32;64,256,1,28,160,32" python tools/wino_bench.py 2>&1 | tail -30
python bench.py --steps 12 --warmup 4 > gpurun_out/b_fold4.json 2> gpurun_out/b_fold4.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_fold4.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.get('other_configs',{}).items()})
PY
