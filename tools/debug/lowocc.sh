# kernel traces of the C3 / C5 / ResNet-50 steps: dispatches with few workgroups and long durations (the finalizer / finish-kernel class of problem)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
for cfg in "repvgg 64 256" "tinyvit 32 128" "resnet50 32 128"; do
  set -- $cfg
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lowocc_$1 -o t -- python3 $R/tools/step_bench.py $1 $2 $3 2 4 > $R/gpurun_out/lowocc_$1.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,collections,re,glob
for net in ('repvgg','tinyvit','resnet50'):
    f=glob.glob(f'gpurun_out/lowocc_{net}/**/*kernel_trace.csv',recursive=True)
    if not f: print(net,'no trace'); continue
    d=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        n=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','').replace('suhpe::','')
        wgs=(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))*(int(r['Grid_Size_Y'])//int(r['Workgroup_Size_Y']))*(int(r['Grid_Size_Z'])//int(r['Workgroup_Size_Z']))
        d[(n,wgs)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    rows=[]
    for (n,w),v in d.items():
        v.sort()
        if w<160 and v[len(v)//2]>12: rows.append((sum(v)/6/1e3,n,w,len(v)/6,v[len(v)//2]))
    print('==',net)
    for r in sorted(rows,reverse=True)[:14]: print("  %.3f ms/step  %-62s wgs=%d x%.1f med %.1f us"%r)
PY
