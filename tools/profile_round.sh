#!/bin/bash
# One profiling pass of a round on the GPU box (run from the repo root through gpurun):
#   bash tools/profile_round.sh r02
# writes under gpurun_out/prof_<tag>/: the rocprofv3 --kernel-trace --stats summaries of the C2 bench (single stream), of the C3
# and C5 steps, a two-stream kernel trace of C2 for tools/timeline.py, and the FETCH_SIZE / WRITE_SIZE counter passes of two C2
# steps (separate --pmc runs, MI355X_MICROARCH.md "HBM"); tools/pmc_traffic.py turns the two counter passes into pmc_traffic.json.
set -u
TAG=${1:-r02}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_serial -o s -- python3 $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-other-configs --serial > $OUT/c2_serial_bench.json 2> $OUT/c2_serial.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/c2_streams -o s -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > $OUT/c2_streams_bench.json 2> $OUT/c2_streams.err
export SUHPE_OVERLAP=0 SUHPE_WGRAD_OVERLAP=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_serial -o s -- python3 $R/tools/step_bench.py repvgg 64 256 2 6 > $OUT/c3_serial.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_serial -o s -- python3 $R/tools/step_bench.py tinyvit 32 128 2 6 > $OUT/c5_serial.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/step_bench.py effinetv2 32 128 1 2 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/step_bench.py effinetv2 32 128 1 2 > $OUT/pmc_write.log 2>&1
if [ "${PMC_OTHERS:-1}" = "1" ]; then      # the same two passes for C3 and C5 (bench.py's other_configs.*.hbm_traffic_ratio)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_repvgg -- python3 $R/tools/step_bench.py repvgg 64 256 1 2 > $OUT/pmc_fetch_repvgg.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_repvgg -- python3 $R/tools/step_bench.py repvgg 64 256 1 2 > $OUT/pmc_write_repvgg.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_tinyvit -- python3 $R/tools/step_bench.py tinyvit 32 128 1 2 > $OUT/pmc_fetch_tinyvit.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_tinyvit -- python3 $R/tools/step_bench.py tinyvit 32 128 1 2 > $OUT/pmc_write_tinyvit.log 2>&1
fi
unset SUHPE_OVERLAP SUHPE_WGRAD_OVERLAP
cd $R
python3 tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_traffic.json "effinetv2 32+128" > $OUT/pmc_traffic.log 2>&1
if [ "${PMC_OTHERS:-1}" = "1" ]; then
python3 tools/pmc_traffic.py $OUT/pmc_fetch_repvgg $OUT/pmc_write_repvgg $OUT/pmc_traffic_repvgg.json "repvgg 64+256" >> $OUT/pmc_traffic.log 2>&1
python3 tools/pmc_traffic.py $OUT/pmc_fetch_tinyvit $OUT/pmc_write_tinyvit $OUT/pmc_traffic_tinyvit.json "tinyvit 32+128" >> $OUT/pmc_traffic.log 2>&1
fi
python3 tools/timeline.py $(ls $OUT/c2_streams/*kernel_trace.csv | head -1) 3 > $OUT/c2_streams_timeline.txt 2>&1
# keep the merge small: the raw traces stay on the box, the summaries travel
for d in c2_serial c3_serial c5_serial; do cp $OUT/$d/*kernel_stats.csv $OUT/${d}_kernel_stats.csv 2>/dev/null; done
mkdir -p $OUT/pmc && cp $OUT/pmc_fetch/*/*counter_collection.csv $OUT/pmc/fetch_size_counter_collection.csv 2>/dev/null; cp $OUT/pmc_write/*/*counter_collection.csv $OUT/pmc/write_size_counter_collection.csv 2>/dev/null
rm -rf $OUT/c2_serial $OUT/c2_streams $OUT/c3_serial $OUT/c5_serial $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_repvgg $OUT/pmc_write_repvgg $OUT/pmc_fetch_tinyvit $OUT/pmc_write_tinyvit
ls -la $OUT
