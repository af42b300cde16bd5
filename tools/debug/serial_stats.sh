#!/bin/bash
# per-kernel totals of the C2 step on one stream:  bash tools/debug/serial_stats.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-cur}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_serial -o s -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --serial > $R/gpurun_out/${TAG}_serial.json 2>/dev/null
cd $R; cp gpurun_out/${TAG}_serial/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv; rm -rf gpurun_out/${TAG}_serial
