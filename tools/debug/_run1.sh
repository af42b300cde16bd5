cd $GRAFT_REPO_ROOT
for i in 1 2; do python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1; done
python tools/step_bench.py repvgg 64 256 3 8 2>&1 | tail -1
python tools/step_bench.py tinyvit 32 128 3 8 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_segments.py tests/test_gpu_backbones.py tests/test_gpu_attention_recompute.py tests/test_gpu_guard_bands.py -x -q 2>&1 | tail -2
