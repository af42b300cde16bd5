cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_guard_bands.py -x -q 2>&1 | tail -2
python tools/step_bench.py repvgg 64 256 3 8 2>&1 | tail -1
python tools/step_bench.py repvgg 64 256 3 8 2>&1 | tail -1
python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
WINO_SHAPES="128,128,1,56,320,64;512,512,1,14,320,64;48,192,1,56,160,32" timeout 600 python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/.*| wgrad wino-direct/wgrad wino-direct/' | cut -c1-110
