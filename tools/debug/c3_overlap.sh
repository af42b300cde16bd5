for i in 1 2; do
python tools/step_bench.py repvgg 64 256 3 8 2>/dev/null | tail -1 | cut -c1-60 | sed 's/^/both         /'
SUHPE_OVERLAP=0 python tools/step_bench.py repvgg 64 256 3 8 2>/dev/null | tail -1 | cut -c1-60 | sed 's/^/wgrad only   /'
SUHPE_WGRAD_OVERLAP=0 python tools/step_bench.py repvgg 64 256 3 8 2>/dev/null | tail -1 | cut -c1-60 | sed 's/^/teacher only /'
SUHPE_OVERLAP=0 SUHPE_WGRAD_OVERLAP=0 python tools/step_bench.py repvgg 64 256 3 8 2>/dev/null | tail -1 | cut -c1-60 | sed 's/^/none         /'
done
