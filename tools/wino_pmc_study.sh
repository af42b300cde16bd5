#!/bin/bash
# Where a Winograd launch spends its time: SQ counters of the last wino_nt_kernel launch of tools/wino_pmc.py, one rocprofv3 --pmc pass per group.
#   bash tools/wino_pmc_study.sh "cin cout groups hw n" [pass]      (run from the repo root through gpurun)
R=$(pwd); cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
shape="$1"; pass=${2:-0}
groups=(
"GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_VALU_MFMA_BUSY_CYCLES"
"SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"
"SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_IFETCH"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"
)
echo "== wino $shape pass $pass"
for g in "${groups[@]}"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/wino_pmc.py $shape $pass > /tmp/pm.log 2>&1
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f:
    print("  (no counter file)"); print(open('/tmp/pm.log').read()[-600:]); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if 'wino_nt' in r['Kernel_Name'] or 'wino_small' in r['Kernel_Name']]
last=max(int(r['Dispatch_Id']) for r in rows)
rr=[r for r in rows if int(r['Dispatch_Id'])==last]
dur=int(rr[0]['End_Timestamp'])-int(rr[0]['Start_Timestamp'])
print(f"  [{rr[0]['Kernel_Name'][:40]}] {dur/1e3:.1f} us  vgpr {rr[0].get('VGPR_Count','?')} lds {rr[0].get('LDS_Block_Size','?')}  " + "  ".join(f"{r['Counter_Name']}={float(r['Counter_Value']):.4g}" for r in rr))
PY
done
