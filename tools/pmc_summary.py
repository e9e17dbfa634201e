"""Per-kernel average of every counter in a rocprofv3 counter_collection.csv.  usage: pmc_summary.py <csv> [filter]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:70]
    if flt in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k, {c: (len(v), round(sum(v) / len(v), 1)) for c, v in d.items()})
