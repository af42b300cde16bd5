#!/bin/bash
# Clock, MFMA-pipe and LDS activity of the NT GEMM tiles on pointwise shapes of the C2 step (run from the repo root through gpurun):
#   bash tools/gemm_clock_study.sh > gpurun_out/gemm_clock_study.txt
# One rocprofv3 --pmc pass per (shape, tile): GRBM_GUI_ACTIVE (summed over the 8 XCDs: / 8 / duration = the clock the chip held),
# SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / 4 (share of the time a SIMD's matrix pipe was busy; 1.000 on tools/micro/mfma_peak),
# SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES, SQ_LDS_BANK_CONFLICT.
R=$(pwd); cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
# menu indices resolved BY NAME (the menu grows: fixed numbers went stale once)
TILES=$(python3 - <<'PY'
import ctypes
from semiuhpe_amd import _lib
lib = _lib.load(); lib.suhpe_conv2d_tile_name.restype = ctypes.c_char_p
names = [lib.suhpe_conv2d_tile_name(0, i).decode() for i in range(lib.suhpe_conv2d_tile_count(0))]
print(" ".join(str(names.index(n)) for n in ("igemm_nt_kernel<64,64,32,32,4>", "igemm_nt_kernel<128,64,32,32,4>", "gemm_nt_resident<128,2,16>", "gemm_nt_resident<64,2,8>", "gemm_nt_dma<128,128>", "gemm_nt_dma<256,64>")))
PY
)
# shapes "M N K": the two VERDICT r4 names (160 -> 960 @ 14^2, 256 -> 1536 @ 7^2), 128 -> 512, a narrow-output stream, a long reduction
SHAPE_LIST=("31360 960 160" "7840 1536 256" "31360 512 128" "501760 64 192" "31360 960 2048")
for shape in "${SHAPE_LIST[@]}"; do
  for tile in $TILES; do
    rm -rf /tmp/pm; timeout 180 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/gemm_pmc.py $shape 0 $tile > /dev/null 2>&1
    python3 - "$shape" $tile <<'PY'
import csv,glob,sys
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name']]
last=max(int(r['Dispatch_Id']) for r in rows)
d={r['Counter_Name']:float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last}
rr=[r for r in rows if int(r['Dispatch_Id'])==last][0]
dur=int(rr['End_Timestamp'])-int(rr['Start_Timestamp'])
m,n,k=[int(v) for v in sys.argv[1].split()]
name=rr['Kernel_Name'].replace('void suhpe::','')[:44]
print(f"M N K {sys.argv[1]:16s} tile {sys.argv[2]:>2s} {name:44s} {dur/1e3:7.1f} us {2*m*n*k/dur/1e3:6.1f} TF/s  clock {d['GRBM_GUI_ACTIVE']/8/(dur/1e3):5.0f} MHz  mfma_busy {d['SQ_VALU_MFMA_BUSY_CYCLES']/d['SQ_BUSY_CU_CYCLES']/4:.3f}  lds_active {d['SQ_LDS_IDX_ACTIVE']/d['SQ_BUSY_CU_CYCLES']:.3f}  bank_conflicts {d['SQ_LDS_BANK_CONFLICT']/max(1,d['SQ_LDS_IDX_ACTIVE']):.3f}")
PY
  done
done
rm -rf /tmp/pm; rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d /tmp/pm -- python3 $R/tools/micro/run_mfma_peak.py > /tmp/mp.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/pm/**/*counter_collection.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
last=max(int(r['Dispatch_Id']) for r in rows)
d={r['Counter_Name']:float(r['Counter_Value']) for r in rows if int(r['Dispatch_Id'])==last}
rr=[r for r in rows if int(r['Dispatch_Id'])==last][0]
dur=int(rr['End_Timestamp'])-int(rr['Start_Timestamp'])
print(f"register-only MFMA loop ({rr['Kernel_Name'][:30]}): clock {d['GRBM_GUI_ACTIVE']/8/(dur/1e3):5.0f} MHz  mfma_busy {d['SQ_VALU_MFMA_BUSY_CYCLES']/d['SQ_BUSY_CU_CYCLES']/4:.3f}")
PY
grep -i "TF" /tmp/mp.log | tail -2
