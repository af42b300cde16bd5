"""Average the counters of a rocprofv3 --pmc run per kernel:  python tools/pmc_summary.py <dir> [name filter]"""
import csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if flt not in name:
            continue
        a = agg.setdefault((name[:70], r["Counter_Name"]), [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
for (name, c), (v, n) in sorted(agg.items()):
    print(f"{name:70s} {c:32s} avg {v/n:16.1f} over {n}")
