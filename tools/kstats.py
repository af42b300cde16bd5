"""Print rocprofv3 --stats kernel rows matching any of the given substrings:   python tools/kstats.py <dir> [substr ...]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
pats = sys.argv[2:]
for r in csv.DictReader(open(f)):
    if not pats or any(p in r["Name"] for p in pats):
        print("%9.1f us x%5s  min %8.1f max %8.1f  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                         r["Name"].replace("void suhpe::", "")[:100]))
