cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
cp gpurun_out/prof_r05/pmc_traffic.json profiles/pmc_traffic.json 2>/dev/null
cp gpurun_out/prof_r05/pmc_traffic_repvgg.json profiles/pmc_traffic_repvgg.json 2>/dev/null
cp gpurun_out/prof_r05/pmc_traffic_tinyvit.json profiles/pmc_traffic_tinyvit.json 2>/dev/null
python bench.py > gpurun_out/bench_default_r05.json 2> gpurun_out/bench_default_r05.err
tail -c 600 gpurun_out/bench_default_r05.json
