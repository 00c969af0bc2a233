"""Aggregate a rocprofv3 counter_collection.csv per kernel: sum of each counter over all dispatches."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[(r["Kernel_Name"], r["Counter_Name"])] += 1
names = sorted({c for v in agg.values() for c in v})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches"] + names)
for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
    w.writerow([k, max(calls[(k, c)] for c in v)] + ["%.6g" % v.get(c, 0.0) for c in names])
