python -m pytest tests/test_gpu_winograd.py -m gpu -x -q -k "float64 or refus" 2>&1 | tail -4
WINO_SHAPES="24,24,1,112,160,32;24,24,1,112,128,32" python tools/wino_bench.py 2>&1 | tail -2
