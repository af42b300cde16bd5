#!/bin/bash
# Vector-ALU against MFMA activity of the Winograd kernels (DESIGN.md 3.1 "Round 5" (7): fp32 MFMAs run on the vector ALU, so a kernel's time on a SIMD is
# MFMA cycles + ~4.5 cycles per other vector instruction): two rocprofv3 --pmc passes per kernel (SQ activity / instruction counts), the last launch of each.
#   bash tools/debug/valu_mfma_pmc.sh        (from the repo root through gpurun; writes gpurun_out/r05_wino_bench.txt and gpurun_out/r05_valu_mfma_pmc.txt)
# Do NOT add TA_* counters: that group hung rocprofv3 on this pool.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R" || exit 1
mkdir -p gpurun_out
python tools/wino_bench.py all 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_wino_bench.txt
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
groups=(
"GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
"SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU"
)
for tgt in "wino_pmc.py 24 24 1 112 160 0|wino_small_kernel" "wino_pmc.py 24 24 1 112 160 1|wino_small_kernel" "debug/wgrad_small_prof.py 24 24 1 112 160|wino_small_wgrad" "wino_pmc.py 48 192 1 56 160 0|wino_nt_kernel" "debug/wgrad_small_prof.py 128 128 1 56 320|wino_wgrad_kernel"; do
  cmd=${tgt%%|*}; pat=${tgt##*|}
  echo "== $cmd  [$pat]"
  for g in "${groups[@]}"; do
    rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/$cmd > /tmp/pm.log 2>&1
    PAT=$pat python3 - <<'PY'
import csv,glob,os
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f:
    print("  (no counter file)"); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if os.environ['PAT'] in r['Kernel_Name']]
if not rows:
    print('  (no matching kernel)'); raise SystemExit
last=max(int(r['Dispatch_Id']) for r in rows)
rr=[r for r in rows if int(r['Dispatch_Id'])==last]
dur=int(rr[0]['End_Timestamp'])-int(rr[0]['Start_Timestamp'])
print(f"  [{rr[0]['Kernel_Name'][:44]}] {dur/1e3:.1f} us  " + "  ".join(f"{r['Counter_Name']}={float(r['Counter_Value']):.4g}" for r in rr))
PY
  done
done > $R/gpurun_out/r05_valu_mfma_pmc.txt 2>&1
tail -12 $R/gpurun_out/r05_valu_mfma_pmc.txt | cut -c1-260
