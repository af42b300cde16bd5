cd $GRAFT_REPO_ROOT
for i in 1 2; do
echo "== new DB (small fwd/dgrad/wgrad)"; python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
echo "== small wgrad off"; SUHPE_WINO_SMALL_WGRAD=0 SUHPE_TILE_DB=$PWD/tools/debug/data/tiles_wgrad_direct.json python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
echo "== old DB"; SUHPE_TILE_DB=$PWD/tools/debug/data/tiles_pre_small.json python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
done > gpurun_out/step_ab.txt 2>&1
cat gpurun_out/step_ab.txt
