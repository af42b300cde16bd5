#!/bin/bash
# which small runtime kernels (copyBuffer / fill / elementwise) run INSIDE a steady-state step?  (kernel trace split at adam_kernel)
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$R
rocprofv3 --kernel-trace --output-format csv -d /tmp/cc -o s -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs > /dev/null 2>/tmp/cc.err
python3 - <<'PY'
import csv, glob, re, collections
f = glob.glob('/tmp/cc/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
cuts = [i for i, r in enumerate(rows) if "adam_kernel" in r[1]]
print(len(rows), "dispatches,", len(cuts), "adam launches")
for s in range(max(1, len(cuts) - 2), len(cuts)):
    seg = rows[cuts[s - 1] + 1: cuts[s] + 1]
    c = collections.Counter(re.sub(r"\(.*", "", n)[:60] for _, n in seg if "suhpe" not in n)
    print("step", s, "launches", len(seg), "non-library kernels:", dict(c))
PY
