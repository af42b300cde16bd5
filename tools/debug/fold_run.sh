set -x
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "long_record or fused_bn_statistics or segment" 2>&1 | tail -5
python -m pytest tests/test_gpu_winograd.py -m gpu -x -q -k "network_parity" 2>&1 | tail -15
python bench.py --steps 12 --warmup 4 > gpurun_out/b_fold3.json 2> gpurun_out/b_fold3.err; tail -c 600 gpurun_out/b_fold3.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/b_fold3.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:v.get('ms_per_step') for k,v in d.get('other_configs',{}).items()})
PY
