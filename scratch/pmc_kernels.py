"""Per-kernel mean of every counter in a rocprofv3 counter_collection.csv (kernels matching argv[2])."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]: continue
    k = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
    a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, cs in agg.items():
    print(k, {c: round(v[1] / v[0]) for c, v in cs.items()})
