"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel: python tools/pmc_sq.py <dir> -> table on stdout."""
import csv, glob, os, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[(k, row["Counter_Name"])] += 1
names = sorted({c for v in agg.values() for c in v})
print("kernel," + ",".join(names) + ",dispatches")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    print(k + "," + ",".join("%.4g" % v.get(c, 0) for c in names) + ",%d" % cnt[(k, names[0])])
