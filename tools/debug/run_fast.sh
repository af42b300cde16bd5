cd $GRAFT_REPO_ROOT
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/b_fast.json 2> gpurun_out/b_fast.err
python -c "
import json; d=json.load(open('gpurun_out/b_fast.json')); print('ms/step', d['ms_per_step'], 'gemm', d['roofline'].get('gemm_ms_per_step'))"
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_effnet.py tests/test_gpu_segments.py -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fast_serial -o s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --serial > $GRAFT_REPO_ROOT/gpurun_out/fast_serial.json 2>/dev/null
cd $GRAFT_REPO_ROOT; cp gpurun_out/fast_serial/*kernel_stats.csv gpurun_out/fast_serial_kernel_stats.csv; rm -rf gpurun_out/fast_serial
