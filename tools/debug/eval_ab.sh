python -m pytest tests/test_gpu_winograd.py -m gpu -x -q -k "eval_epilogue or refused or float64" 2>&1 | tail -4
for net in repvgg effinetv2 resnet50; do
python tools/eval_bench.py $net 256 2>&1 | tail -1
SUHPE_WINOGRAD=0 python tools/eval_bench.py $net 256 2>&1 | tail -1 | sed 's/^/   (SUHPE_WINOGRAD=0) /'
done
