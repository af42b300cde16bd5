cd $GRAFT_REPO_ROOT
export WINO_SHAPES="24,24,1,10,5,2;24,32,1,12,4,3;24,24,1,112,160,32"
timeout 600 python tools/wino_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-700 > gpurun_out/wsmall_on.txt
cat gpurun_out/wsmall_on.txt
timeout 900 python -m pytest tests/test_gpu_winograd.py tests/test_gpu_guard_bands.py -x -q 2>&1 | tail -3
