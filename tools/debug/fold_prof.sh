#!/bin/bash
# per-kernel comparison of SUHPE_FUSE_EXPAND_DW=0/1 on the C2 bench (serial streams): run from the repo root through gpurun
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
for v in 0 1; do
export SUHPE_FUSE_EXPAND_DW=$v
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$v -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --serial > /dev/null 2>/tmp/prof$v.err
python3 - <<PY
import csv,glob,re,collections
f=glob.glob('/tmp/prof$v/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    n=re.sub(r'suhpe::','',r['Name']); n=re.sub(r'\(.*','',n)
    agg[n[:70]][0]+=int(r['Calls']); agg[n[:70]][1]+=float(r['TotalDurationNs'])
tot=sum(v[1] for v in agg.values())
print("fold=$v kernel ms/step", round(tot/9/1e6,3), "launches/step", sum(v[0] for v in agg.values())/9)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1]):
    if any(s in k for s in ('dw3x3','bn_act_fwd','bn_act_bwd','dw_wgrad','se_bn','bn_act_pool','bn_act_scale')):
        print(f"  {k:70s} {v[0]/9:7.1f} {v[1]/9/1e6:8.3f}")
PY
done
