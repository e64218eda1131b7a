"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: launches, mean counter value per launch."""
import csv, sys, re, collections
rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:100]
    key = (name, r["Counter_Name"])
    agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
print("kernel,counter,launches,mean_per_launch,total")
for (name, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:200]:
    print(f'"{name}",{c},{n},{tot/n:.1f},{tot:.1f}')
