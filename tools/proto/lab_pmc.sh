#!/bin/bash
# PMC view of every kernel the GEMM lab launches: per kernel name, median over its dispatches.
#   bash tools/proto/lab_pmc.sh "<lab args>"    (from the repo root, through gpurun)
R=$(pwd); cd /tmp && export TMPDIR=/tmp
groups=(
"GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_MFMA"
)
for g in "${groups[@]}"; do
  rm -rf /tmp/pm; timeout 200 rocprofv3 --pmc $g --kernel-trace --output-format csv -d /tmp/pm -- $R/tools/proto/bin/gemm_lab $1 > /tmp/pm.log 2>&1
  python3 - <<'PY'
import csv,glob,collections,statistics
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
if not f:
    print("(no counter file)"); print(open('/tmp/pm.log').read()[-2000:]); raise SystemExit
per=collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    n=r['Kernel_Name']
    if 'gemm' not in n: continue
    d=per.setdefault(n,{}).setdefault(int(r['Dispatch_Id']),{})
    d[r['Counter_Name']]=float(r['Counter_Value'])
    d['dur']=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    d['vgpr']=r.get('VGPR_Count','?'); d['agpr']=r.get('Accum_VGPR_Count','?'); d['lds']=r.get('LDS_Block_Size','?')
for n,ds in per.items():
    ks=[k for k in next(iter(ds.values())) if k not in('vgpr','agpr','lds')]
    med={k:statistics.median(v[k] for v in ds.values()) for k in ks}
    v0=next(iter(ds.values()))
    cyc=med['GRBM_GUI_ACTIVE']/8
    line=f"{n.replace('void ','').replace('suhpe::','')[:52]:52s} {med['dur']/1e3:7.1f}us clk {cyc/med['dur']*1e3:5.0f}MHz cu_busy {med['SQ_BUSY_CU_CYCLES']/256/cyc:.2f} mfma/dur {med['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc:.3f} mfma/cu_busy {med['SQ_VALU_MFMA_BUSY_CYCLES']/4/med['SQ_BUSY_CU_CYCLES']:.3f}"
    wc=med['SQ_WAVE_CYCLES']
    line+=f" | wave: parked {med['SQ_WAIT_ANY']/wc:.2f} issue-wait {med['SQ_WAIT_INST_ANY']/wc:.2f} active {med['SQ_ACTIVE_INST_ANY']/wc:.2f} waves {med['SQ_WAVES']:.0f} life {wc*4/med['SQ_WAVES']/1e3:.1f}kcyc vgpr {v0['vgpr']}+{v0['agpr']} lds {v0['lds']}"
    print(line)
PY
done
