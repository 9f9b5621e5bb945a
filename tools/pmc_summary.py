"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel name: launches, mean counter value."""
import csv, sys, collections
path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
with open(path) as f:
    r = csv.DictReader(f)
    for row in r:
        name = row.get("Kernel_Name") or row.get("Kernel Name")
        c = row.get("Counter_Name"); v = float(row.get("Counter_Value") or 0)
        a = agg[name][c]; a[0] += 1; a[1] += v
print("kernel,counter,launches,mean_value,total_value")
for name, cs in sorted(agg.items(), key=lambda kv: -sum(x[1] for x in kv[1].values())):
    for c, (n, tot) in cs.items():
        print(f"\"{name[:90]}\",{c},{n},{tot/n:.3f},{tot:.1f}")
