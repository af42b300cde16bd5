#!/bin/bash
# Everything profiles/ needs for round 6, in one gpurun call from the repo root:  bash tools/round6_profiles.sh [tag = r06]
# (profile_round.sh: kernel stats C2 / C3 / C5, two-stream timeline, FETCH_SIZE / WRITE_SIZE passes; then the per-block ledger from the raw counter
#  files, the HBM-bound kernel table, the in-step GEMM shape list, and the default bench line)
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
bash tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
OUT=gpurun_out/prof_$TAG
python3 tools/block_ledger.py $OUT/pmc/fetch_size_counter_collection.csv $OUT/pmc/write_size_counter_collection.csv 10 > $OUT/block_ledger.txt 2>&1
python3 tools/ewbench.py > $OUT/ewbench.txt 2>&1
SERIAL=1 python3 tools/shape_profile.py effinetv2 > $OUT/gemm_shapes_serial.txt 2>&1
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 400 $OUT/bench_default.json
ls -la $OUT
