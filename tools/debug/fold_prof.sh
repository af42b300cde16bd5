cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace -d gpurun_out/foldprof -o f --output-format csv -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/foldprof.log 2>&1
python3 - <<'PY'
import csv,collections,glob
f=glob.glob('gpurun_out/foldprof/**/*kernel_trace.csv',recursive=True)[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name']
    for key in ('bn_records_fold','bn_stats_finalize_folded','bn_stats_finalize2','bn_stats_finalize('):
        if key in n:
            d[(key,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),int(r['Grid_Size_Y']))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()):
    v.sort(); print(k,len(v),'med %.1f min %.1f'%(v[len(v)//2],v[0]))
PY
