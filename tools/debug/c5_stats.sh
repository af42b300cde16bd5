#!/bin/bash
# per-kernel totals of the C5 (TinyViT) step on one stream:  bash tools/debug/c5_stats.sh <tag>   -> gpurun_out/<tag>_c5_kernel_stats.csv
TAG=${1:-cur}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R SUHPE_OVERLAP=0 SUHPE_WGRAD_OVERLAP=0
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_c5 -o s -- python3 $R/tools/step_bench.py tinyvit 32 128 2 6 > $R/gpurun_out/${TAG}_c5.log 2>&1
cd $R; cp gpurun_out/${TAG}_c5/*kernel_stats.csv gpurun_out/${TAG}_c5_kernel_stats.csv; rm -rf gpurun_out/${TAG}_c5
grep "attn" gpurun_out/${TAG}_c5_kernel_stats.csv | cut -c1-60,200-
