cd $GRAFT_REPO_ROOT
echo "xf=1 (default)"; python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
echo "xf=2"; SUHPE_XF_FUSION=2 python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
echo "xf=0"; SUHPE_XF_FUSION=0 python tools/step_bench.py effinetv2 32 128 5 20 2>&1 | tail -1
