#!/bin/bash
# SQ / TCP counters of the LAST launch of a kernel (name substring) in any python tool, one rocprofv3 --pmc pass per counter group:
#   bash tools/kernel_pmc_study.sh <kernel substring> tools/attnbench.py 160 14      (run from the repo root through gpurun)
R=$(pwd); cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
kern="$1"; shift
groups=(
"GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_VALU_MFMA_BUSY_CYCLES"
"SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"
"SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_IFETCH"
"SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"
)
echo "== $kern :: $*"
for g in "${groups[@]}"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/"$@" > /tmp/pm.log 2>&1
  KERN="$kern" python3 - <<'PY'
import csv,glob,os
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f:
    print("  (no counter file)"); print(open('/tmp/pm.log').read()[-600:]); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if os.environ['KERN'] in r['Kernel_Name']]
if not rows:
    print('  (no matching kernel)'); raise SystemExit
last=max(int(r['Dispatch_Id']) for r in rows)
rr=[r for r in rows if int(r['Dispatch_Id'])==last]
dur=int(rr[0]['End_Timestamp'])-int(rr[0]['Start_Timestamp'])
print(f"  [{rr[0]['Kernel_Name'][:40]}] {dur/1e3:.1f} us  vgpr {rr[0].get('VGPR_Count','?')} lds {rr[0].get('LDS_Block_Size','?')}  " + "  ".join(f"{r['Counter_Name']}={float(r['Counter_Value']):.4g}" for r in rr))
PY
done
