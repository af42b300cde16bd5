#!/bin/bash
# Where a GEMM launch spends its time: SQ / TA / TCP / TCC counters of the last launch of tools/gemm_pmc.py, one rocprofv3 --pmc pass per group.
#   bash tools/gemm_pmc_study.sh "M N K" pass tile [extra env]   (run from the repo root through gpurun; prints one block per call)
R=$(pwd); cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
shape="$1"; pass=${2:-0}; tile=${3:--1}
groups=(
"GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_VALU_MFMA_BUSY_CYCLES"
"SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM"
"SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH"
"TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum"
"TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
"TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum"
"TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum"
"TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum TCC_REQ_sum"
)
for g in "${groups[@]}"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/gemm_pmc.py $shape $pass $tile > /tmp/pm.log 2>&1
  python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f:
    print("  (no counter file)"); raise SystemExit
rows=[r for r in csv.DictReader(open(f[0])) if 'gemm' in r['Kernel_Name'] or 'res_' in r['Kernel_Name']]
last=max(int(r['Dispatch_Id']) for r in rows)
rr=[r for r in rows if int(r['Dispatch_Id'])==last]
dur=int(rr[0]['End_Timestamp'])-int(rr[0]['Start_Timestamp'])
print(f"  [{rr[0]['Kernel_Name'][:60]}] {dur/1e3:.1f} us  vgpr {rr[0].get('VGPR_Count','?')} lds {rr[0].get('LDS_Block_Size','?')}  " + "  ".join(f"{r['Counter_Name']}={float(r['Counter_Value']):.4g}" for r in rr))
PY
done
